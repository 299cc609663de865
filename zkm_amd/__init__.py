"""zkm_amd -- MI355X-native STARK/FRI hot path of the zkMIPS prover.

Thin ctypes binding of the C-ABI library `zkm_amd/csrc/libzkmhip.so` (see include/zkm_hip.h).  The class
and method names mirror the plonky2 / zkm-prover items the library replaces (PolynomialBatch.from_values,
.merkle_tree.cap, get_lde_values, Challenger, prove_single_table; reference prover/src/prover.rs:441-641)
so the parity tests read like the reference's own.  There is NO CPU fallback: if the HIP extension is
missing or no GPU is present, every compute entry point raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_LIB_PATH = os.environ.get("ZKM_HIP_LIB") or os.path.join(_CSRC, "libzkmhip.so")   # (ZKM_HIP_LIB: A/B builds of the same in-tree sources)

P = 0xFFFFFFFF00000001
POSEIDON_COLS = 262
KECCAK_SPONGE_COLS = 470
LOGIC_COLS = 69
KECCAK_COLS = 2431
POSEIDON_SPONGE_COLS = 110
TABLE_POSEIDON, TABLE_LOGIC, TABLE_KECCAK_SPONGE, TABLE_KECCAK, TABLE_MEMORY, TABLE_POSEIDON_SPONGE = 0, 1, 2, 3, 4, 5
TABLE_SHA_EXTEND, TABLE_SHA_EXTEND_SPONGE, TABLE_SHA_COMPRESS, TABLE_SHA_COMPRESS_SPONGE, TABLE_ARITHMETIC, TABLE_CPU = 6, 7, 8, 9, 10, 11
u64p = C.POINTER(C.c_uint64)

EXPORTS = [
    "zkm_ctx_create", "zkm_ctx_destroy", "zkm_ctx_memory", "zkm_ctx_resident_bytes", "zkm_ctx_trim", "zkm_ctx_set_tuning", "zkm_table_enum_index", "zkm_host_alloc", "zkm_host_free",
    "zkm_host_register", "zkm_host_unregister", "zkm_all_stark_ctls", "zkm_all_stark_ctl_table", "zkm_prove_segment", "zkm_prove_segments", "zkm_prove_segments_columns", "zkm_prove_segment_columns", "zkm_ctx_synchronize", "zkm_ctx_stream", "zkm_dev_alloc", "zkm_dev_free",
    "zkm_dev_upload", "zkm_dev_download", "zkm_ntt", "zkm_field_selftest", "zkm_batch_commit_values", "zkm_batch_commit_coeffs", "zkm_batch_commit_columns", "zkm_batch_free",
    "zkm_batch_cap", "zkm_batch_coeffs", "zkm_batch_lde_row", "zkm_batch_lde_rows", "zkm_batch_leaf", "zkm_batch_merkle_path",
    "zkm_batch_digest_layer", "zkm_poseidon_permute_batch", "zkm_keccakf_batch", "zkm_poseidon_trace", "zkm_keccak_sponge_trace", "zkm_keccak_trace", "zkm_logic_trace",
    "zkm_poseidon_sponge_trace", "zkm_poseidon_trace_inputs", "zkm_sha_extend_trace", "zkm_sha_extend_sponge_trace",
    "zkm_sha_compress_trace", "zkm_sha_compress_sponge_trace",
    "zkm_table_width", "zkm_num_lookup_columns", "zkm_challenger_init",
    "zkm_challenger_observe", "zkm_challenger_get", "zkm_challenger_compact", "zkm_standard_config", "zkm_proof_words",
    "zkm_prove_single_table", "zkm_prove_single_tables", "zkm_prove_openings", "zkm_fri_prove", "zkm_fri_proof_words", "zkm_prove_single_table_ctl", "zkm_ctl_data", "zkm_lookup_helper_columns", "zkm_all_proof_words", "zkm_prove_with_traces",
    "zkm_proof_get_layout", "zkm_proof_get_query_layout", "zkm_segment_image_words", "zkm_segment_image_write", "zkm_prove_segment_image",
    "zkm_quotient", "zkm_eval_openings", "zkm_check_constraints", "zkm_profile_enable", "zkm_profile_reset",
    "zkm_profile_count", "zkm_profile_get", "zkm_version",
    "zkm_trace_stage", "zkm_trace_stage_columns", "zkm_segment_stage", "zkm_segment_stage_columns", "zkm_staged_segment_ptrs", "zkm_staged_ptr", "zkm_staged_ready", "zkm_staged_free",
    "zkm_pool_create", "zkm_pool_destroy", "zkm_pool_workers", "zkm_pool_context", "zkm_pool_device", "zkm_pool_set_tuning",
    "zkm_pool_prove_segments", "zkm_pool_prove_segments_columns", "zkm_pool_plan", "zkm_pool_last_assignment",
]


class ZkmError(RuntimeError):
    pass


def build(force=False):
    """Compile libzkmhip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", _CSRC, "-j8", "-s"]
    if force:
        subprocess.check_call(["make", "-C", _CSRC, "clean", "-s"])
    subprocess.check_call(args)
    return _LIB_PATH


class Challenger(C.Structure):
    """plonky2 Challenger<F, PoseidonHash> state (host side)."""
    _fields_ = [("state", C.c_uint64 * 12), ("in_buf", C.c_uint64 * 8), ("out_buf", C.c_uint64 * 8),
                ("n_in", C.c_uint32), ("n_out", C.c_uint32)]


class ProofLayout(C.Structure):
    """zkm_proof_layout (include/zkm_hip.h): word offsets of the fields of one proof blob."""
    _fields_ = [(n, C.c_uint64) for n in ("degree_bits", "trace_cols", "aux_cols", "quotient_polys", "ctl_zs", "cap_height", "fri_layers",
                                          "final_poly_len", "num_queries", "rate_bits", "arity_bits")] + \
               [(n, C.c_size_t) for n in ("total_words", "init_challenger_state", "trace_cap", "aux_cap", "quotient_cap", "local_values",
                                          "next_values", "aux_polys", "aux_polys_next", "ctl_zs_first", "quotient_polys_open",
                                          "commit_phase_merkle_caps", "final_poly", "pow_witness", "query_round_proofs",
                                          "query_round_words")]


class ProofQueryLayout(C.Structure):
    """zkm_proof_query_layout: offsets inside one FRI query round."""
    _fields_ = [("oracle_evals", C.c_size_t * 3), ("oracle_cols", C.c_size_t * 3), ("oracle_siblings", C.c_size_t * 3),
                ("initial_siblings", C.c_size_t), ("layer_evals", C.c_size_t * 16), ("layer_siblings", C.c_size_t * 16),
                ("layer_siblings_count", C.c_size_t * 16)]


class StarkConfig(C.Structure):
    _fields_ = [(n, C.c_uint) for n in
                ("rate_bits", "cap_height", "pow_bits", "num_challenges", "num_queries", "arity_bits", "final_poly_bits")]


class FriPoly(C.Structure):
    """zkm_fri_poly: one polynomial of a FRI batch = (oracle index, column index)."""
    _fields_ = [("oracle", C.c_uint32), ("poly", C.c_uint32)]


class FriBatch(C.Structure):
    """zkm_fri_batch: the polynomials opened at one point (FriBatchInfo, stark.rs:127-148)."""
    _fields_ = [("point", C.c_uint64 * 2), ("polys", C.c_void_p), ("npolys", C.c_size_t)]


def abi_mirrors():
    """Every struct of include/zkm_hip.h -> its Python mirror (a ctypes Structure or a numpy dtype); tests/test_abi.py compares
    size and field offsets of each with what the C compiler says (tools/abi_layout.c)."""
    from . import ctl
    return {"zkm_challenger": Challenger, "zkm_stark_config": StarkConfig, "zkm_proof_layout": ProofLayout,
            "zkm_proof_query_layout": ProofQueryLayout, "zkm_column": ctl.COLUMN_DT, "zkm_colset": ctl.COLSET_DT,
            "zkm_ctl_table": ctl.CtlTableStruct, "zkm_ctl_z": ctl.CTLZ_DT, "zkm_ctl_side": ctl.SIDE_DT,
            "zkm_cross_table_lookup": ctl.CTL_DT, "zkm_table_input": ctl.TableInputStruct, "zkm_fri_poly": FriPoly,
            "zkm_fri_batch": FriBatch}


_lib = None


def load():
    """Load the extension (no GPU needed just to load and resolve symbols)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ZkmError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % _LIB_PATH)
    L = C.CDLL(_LIB_PATH)
    cp, cpp = C.c_void_p, C.POINTER(C.c_void_p)
    err = C.POINTER(C.c_char_p)
    sigs = {
        "zkm_ctx_create": (C.c_int, [C.c_int, cpp, err]),
        "zkm_ctx_destroy": (None, [cp]),
        "zkm_ctx_synchronize": (C.c_int, [cp, err]),
        "zkm_ctx_stream": (cp, [cp]),
        "zkm_dev_alloc": (C.c_int, [cp, C.c_size_t, cpp, err]),
        "zkm_dev_free": (C.c_int, [cp, cp]),
        "zkm_dev_upload": (C.c_int, [cp, cp, cp, C.c_size_t, err]),
        "zkm_dev_download": (C.c_int, [cp, cp, cp, C.c_size_t, err]),
        "zkm_field_selftest": (C.c_int, [cp, u64p, u64p, C.c_size_t, u64p, err]),
        "zkm_ntt": (C.c_int, [cp, cp, C.c_size_t, C.c_uint, C.c_int, C.c_uint64, err]),
        "zkm_batch_commit_values": (C.c_int, [cp, cp, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, cpp, err]),
        "zkm_batch_commit_coeffs": (C.c_int, [cp, cp, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, cpp, err]),
        "zkm_batch_commit_columns": (C.c_int, [cp, C.POINTER(C.c_void_p), C.c_size_t, C.c_uint, C.c_int, C.c_uint, C.c_uint, cpp, err]),
        "zkm_batch_free": (None, [cp]),
        "zkm_batch_cap": (C.c_int, [cp, u64p]),
        "zkm_batch_coeffs": (C.c_int, [cp, cp]),
        "zkm_batch_lde_row": (C.c_int, [cp, C.c_size_t, u64p]),
        "zkm_batch_lde_rows": (C.c_int, [cp, C.c_size_t, C.c_size_t, C.c_size_t, cp]),
        "zkm_batch_leaf": (C.c_int, [cp, C.c_size_t, u64p]),
        "zkm_batch_merkle_path": (C.c_int, [cp, C.c_size_t, u64p]),
        "zkm_batch_digest_layer": (C.c_int, [cp, C.c_uint, u64p]),
        "zkm_poseidon_permute_batch": (C.c_int, [cp, cp, C.c_size_t, err]),
        "zkm_keccakf_batch": (C.c_int, [cp, cp, C.c_size_t, err]),
        "zkm_poseidon_trace": (C.c_int, [cp, C.c_uint64, C.c_size_t, C.c_uint, cp, err]),
        "zkm_keccak_sponge_trace": (C.c_int, [cp, cp, u64p, u64p, C.c_size_t, C.c_uint, cp, C.POINTER(C.c_size_t), err]),
        "zkm_poseidon_sponge_trace": (C.c_int, [cp, cp, u64p, u64p, C.c_size_t, C.c_uint, cp, C.POINTER(C.c_size_t), err]),
        "zkm_poseidon_trace_inputs": (C.c_int, [cp, cp, cp, C.c_size_t, C.c_uint, cp, err]),
        "zkm_sha_extend_trace": (C.c_int, [cp, cp, cp, C.c_size_t, C.c_uint, cp, err]),
        "zkm_sha_extend_sponge_trace": (C.c_int, [cp, cp, cp, C.c_size_t, C.c_uint, cp, err]),
        "zkm_sha_compress_trace": (C.c_int, [cp, cp, cp, cp, C.c_size_t, C.c_uint, cp, err]),
        "zkm_sha_compress_sponge_trace": (C.c_int, [cp, cp, cp, cp, C.c_size_t, C.c_uint, cp, err]),
        "zkm_keccak_trace": (C.c_int, [cp, cp, cp, C.c_size_t, C.c_uint, cp, err]),
        "zkm_ctx_memory": (None, [cp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
        "zkm_ctx_resident_bytes": (C.c_size_t, [cp]),
        "zkm_ctx_trim": (None, [cp]),
        "zkm_ctx_set_tuning": (C.c_int, [cp, C.c_char_p, C.c_uint64, err]),
        "zkm_all_stark_ctls": (C.c_int, [cpp, C.POINTER(C.c_size_t), cpp, C.POINTER(C.c_size_t)]),
        "zkm_all_stark_ctl_table": (cp, [C.c_int]),
        "zkm_prove_segment": (C.c_int, [cp, C.POINTER(StarkConfig), C.POINTER(C.c_void_p), C.POINTER(C.c_uint), u64p, C.c_size_t, u64p,
                                        C.POINTER(C.c_size_t), u64p, err]),
        "zkm_prove_segments": (C.c_int, [cp, C.POINTER(StarkConfig), C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), err]),
        "zkm_prove_segments_columns": (C.c_int, [cp, C.POINTER(StarkConfig), C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                 C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), err]),
        "zkm_prove_segment_columns": (C.c_int, [cp, C.POINTER(StarkConfig), C.POINTER(C.c_void_p), C.POINTER(C.c_uint), u64p, C.c_size_t, u64p,
                                                C.POINTER(C.c_size_t), u64p, err]),
        "zkm_pool_create": (C.c_int, [C.POINTER(C.c_int), C.c_size_t, C.c_size_t, cpp, err]),
        "zkm_pool_destroy": (None, [cp]),
        "zkm_pool_workers": (C.c_size_t, [cp]),
        "zkm_pool_context": (cp, [cp, C.c_size_t]),
        "zkm_pool_device": (C.c_int, [cp, C.c_size_t]),
        "zkm_pool_set_tuning": (C.c_int, [cp, C.c_char_p, C.c_uint64, err]),
        "zkm_pool_prove_segments": (C.c_int, [cp, C.POINTER(StarkConfig), C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), err]),
        "zkm_pool_prove_segments_columns": (C.c_int, [cp, C.POINTER(StarkConfig), C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                      C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), err]),
        "zkm_pool_plan": (C.c_size_t, [C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t), C.c_size_t]),
        "zkm_pool_last_assignment": (C.c_int, [cp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
        "zkm_trace_stage": (C.c_int, [cp, cp, C.c_size_t, C.c_uint, C.c_int, cpp, err]),
        "zkm_trace_stage_columns": (C.c_int, [cp, C.POINTER(C.c_void_p), C.c_size_t, C.c_uint, C.c_int, cpp, err]),
        "zkm_segment_stage": (C.c_int, [cp, C.POINTER(C.c_void_p), C.POINTER(C.c_uint), C.c_int, cpp, err]),
        "zkm_segment_stage_columns": (C.c_int, [cp, C.POINTER(C.c_void_p), C.POINTER(C.c_uint), C.c_int, cpp, err]),
        "zkm_staged_segment_ptrs": (C.c_int, [cp, C.POINTER(C.c_void_p)]),
        "zkm_staged_ptr": (cp, [cp]),
        "zkm_staged_ready": (C.c_int, [cp, C.c_int]),
        "zkm_staged_free": (None, [cp]),
        "zkm_host_alloc": (C.c_int, [cp, C.c_size_t, cpp, err]),
        "zkm_host_free": (C.c_int, [cp, cp]),
        "zkm_host_register": (C.c_int, [cp, cp, C.c_size_t, err]),
        "zkm_host_unregister": (C.c_int, [cp, cp]),
        "zkm_table_enum_index": (C.c_int, [C.c_int]),
        "zkm_logic_trace": (C.c_int, [cp, cp, C.c_size_t, C.c_uint, cp, err]),
        "zkm_table_width": (C.c_size_t, [C.c_int]),
        "zkm_challenger_init": (None, [C.POINTER(Challenger)]),
        "zkm_challenger_observe": (None, [C.POINTER(Challenger), u64p, C.c_size_t]),
        "zkm_challenger_get": (C.c_uint64, [C.POINTER(Challenger)]),
        "zkm_challenger_compact": (None, [C.POINTER(Challenger), u64p]),
        "zkm_standard_config": (None, [C.POINTER(StarkConfig)]),
        "zkm_proof_words": (C.c_size_t, [C.POINTER(StarkConfig), C.c_uint, C.c_size_t, C.c_size_t, C.c_size_t]),
        "zkm_prove_single_table": (C.c_int, [cp, C.c_int, C.POINTER(StarkConfig), cp, C.c_size_t, C.c_uint, cp, cp, C.c_size_t,
                                             C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(Challenger), u64p, err]),
        "zkm_prove_single_tables": (C.c_int, [cp, C.c_int, C.POINTER(StarkConfig), C.c_size_t, C.POINTER(C.c_void_p), C.c_size_t, C.c_uint,
                                              C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_void_p), err]),
        "zkm_fri_proof_words": (C.c_size_t, [C.POINTER(StarkConfig), C.c_uint, C.POINTER(C.c_size_t), C.c_size_t]),
        "zkm_fri_prove": (C.c_int, [cp, C.POINTER(StarkConfig), cpp, C.c_size_t, cp, C.c_size_t, C.POINTER(Challenger), u64p, err]),
        "zkm_prove_openings": (C.c_int, [cp, C.POINTER(StarkConfig), cp, cp, cp, C.c_size_t, C.POINTER(Challenger), u64p, err]),
        "zkm_prove_single_table_ctl": (C.c_int, [cp, C.c_int, C.POINTER(StarkConfig), cp, C.c_size_t, C.c_uint, cp, cp, C.c_size_t,
                                                 cp, cp, cp, C.c_size_t, u64p, C.POINTER(Challenger), u64p, err]),
        "zkm_num_lookup_columns": (C.c_size_t, [C.c_int, C.POINTER(StarkConfig)]),
        "zkm_ctl_data": (C.c_int, [cp, cp, cp, cp, C.c_size_t, cp, C.c_size_t, C.c_uint, cp, err]),
        "zkm_lookup_helper_columns": (C.c_int, [cp, cp, cp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64, cp, C.c_size_t, C.c_uint, cp,
                                                err]),
        "zkm_all_proof_words": (C.c_size_t, [C.POINTER(StarkConfig), cp, C.c_size_t, cp, cp, C.c_size_t, C.POINTER(C.c_size_t)]),
        "zkm_prove_with_traces": (C.c_int, [cp, C.POINTER(StarkConfig), cp, C.c_size_t, cp, cp, C.c_size_t, u64p, C.c_size_t, u64p,
                                            u64p, err]),
        "zkm_proof_get_layout": (C.c_int, [u64p, C.POINTER(ProofLayout)]),
        "zkm_proof_get_query_layout": (C.c_int, [u64p, C.POINTER(ProofQueryLayout)]),
        "zkm_segment_image_words": (C.c_size_t, [cp, C.c_size_t, cp, cp, C.c_size_t, C.c_size_t]),
        "zkm_segment_image_write": (C.c_int, [cp, C.c_size_t, cp, cp, C.c_size_t, u64p, C.c_size_t, u64p, err]),
        "zkm_prove_segment_image": (C.c_int, [cp, C.POINTER(StarkConfig), u64p, C.c_size_t, u64p, C.POINTER(C.c_size_t),
                                              C.POINTER(C.c_size_t), u64p, err]),
        "zkm_quotient": (C.c_int, [cp, C.c_int, cp, cp, C.POINTER(C.c_uint32), C.c_size_t, u64p, C.c_size_t, cp, err]),
        "zkm_eval_openings": (C.c_int, [cp, cp, u64p, u64p, err]),
        "zkm_check_constraints": (C.c_int, [cp, C.c_int, C.POINTER(StarkConfig), cp, C.c_size_t, C.c_uint, cp, C.c_size_t, cp, cp, cp, C.c_size_t,
                                            u64p, u64p, C.c_size_t, u64p, err]),
        "zkm_profile_enable": (None, [cp, C.c_int]),
        "zkm_profile_reset": (None, [cp]),
        "zkm_profile_count": (C.c_size_t, [cp]),
        "zkm_profile_get": (C.c_int, [cp, C.c_size_t, C.POINTER(C.c_char_p), u64p, C.POINTER(C.c_double)]),
        "zkm_version": (C.c_char_p, []),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _np_ptr(a):
    assert isinstance(a, np.ndarray) and a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _check(rc, err):
    if rc != 0:
        msg = err.value.decode() if err.value else "error %d" % rc
        # message was malloc'd by the library (reference convention: caller frees)
        raise ZkmError(msg)


class DeviceBuffer:
    """A chunk of HBM owned by a Context (uint64 words)."""

    def __init__(self, ctx, words):
        self.ctx, self.words = ctx, words
        p = C.c_void_p()
        err = C.c_char_p()
        _check(ctx.L.zkm_dev_alloc(ctx.h, words * 8, C.byref(p), C.byref(err)), err)
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        assert arr.size <= self.words
        err = C.c_char_p()
        _check(self.ctx.L.zkm_dev_upload(self.ctx.h, self.ptr, _np_ptr(arr), arr.size * 8, C.byref(err)), err)
        return self

    def download(self, words=None, offset=0):
        words = self.words - offset if words is None else words
        out = np.empty(words, dtype=np.uint64)
        err = C.c_char_p()
        _check(self.ctx.L.zkm_dev_download(self.ctx.h, _np_ptr(out), self.ptr + 8 * offset, words * 8, C.byref(err)), err)
        return out

    def free(self):
        if self.ptr:
            self.ctx.L.zkm_dev_free(self.ctx.h, self.ptr)
            self.ptr = None


class StagedTrace:
    """zkm_staged: a host matrix on its way into HBM behind the context's current work (include/zkm_hip.h "staged traces").  Pass it
    wherever a prove call of the SAME context takes a trace; free() after that call has returned."""

    def __init__(self, ctx, handle, words):
        self.ctx, self.h, self.words = ctx, handle, words

    @property
    def ptr(self):
        p = self.ctx.L.zkm_staged_ptr(self.h)
        if not p:
            raise ZkmError("zkm_staged_ptr: the upload could not be ordered before the compute stream")
        return p

    def tables(self):
        """A staged SEGMENT's twelve device matrices (integers), ordered behind the upload: traces[s] of prove_segments."""
        out = (C.c_void_p * 12)()
        if self.ctx.L.zkm_staged_segment_ptrs(self.h, out) != 0:
            raise ZkmError("zkm_staged_segment_ptrs: not a staged segment, or the upload could not be ordered before the compute stream")
        return [int(out[t]) for t in range(12)]

    def ready(self, wait=False):
        r = self.ctx.L.zkm_staged_ready(self.h, 1 if wait else 0)
        if r < 0:
            raise ZkmError("zkm_staged_ready: runtime error")
        return bool(r)

    def free(self):
        if self.h:
            self.ctx.L.zkm_staged_free(self.h)
            self.h = None


def _data_ptr(x):
    """Accept numpy host arrays, DeviceBuffers, staged traces, raw integer device pointers or torch CUDA tensors."""
    if isinstance(x, (DeviceBuffer, StagedTrace)):
        return C.c_void_p(x.ptr)
    if isinstance(x, np.ndarray):
        return _np_ptr(x)
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    raise TypeError("unsupported buffer type %r" % type(x))


class _marshal_segments:
    """The argument arrays of zkm_prove_segments[_columns] / zkm_pool_prove_segments[_columns] for segments = list of (traces, log_ns,
    public_values): traces[t] either one block per table or a list of per-column arrays (each its own allocation, like
    Vec<PolynomialValues<F>>).  Output buffers are sized with the sizing pass of zkm_prove_segment[_columns] (no context needed)."""

    def __init__(self, L, segments, cfg):
        K = len(segments)
        self.keep, self.ptr_arrays, self.log_arrays, self.pubs, self.sizes, self.col_arrays = [], [], [], [], [], []
        err = C.c_char_p()
        self.by_columns = all(isinstance(t, (list, tuple)) for seg in segments for t in seg[0])
        for traces, log_ns, public_values in segments:
            assert len(traces) == 12 and len(log_ns) == 12
            if self.by_columns:
                k = [[c if isinstance(c, (DeviceBuffer, StagedTrace)) else np.ascontiguousarray(c, dtype=np.uint64) for c in t] for t in traces]
                cols = [(C.c_void_p * len(t))(*[_data_ptr(c).value for c in t]) for t in k]
                self.col_arrays.append(cols)
                self.keep.append(k)
                self.ptr_arrays.append((C.c_void_p * 12)(*[C.addressof(c) for c in cols]))
            else:
                k = [t if isinstance(t, (DeviceBuffer, StagedTrace, int)) else np.ascontiguousarray(t, dtype=np.uint64) for t in traces]
                self.keep.append(k)
                self.ptr_arrays.append((C.c_void_p * 12)(*[_data_ptr(t).value for t in k]))
            self.log_arrays.append((C.c_uint * 12)(*[int(x) for x in log_ns]))
            self.pubs.append(np.ascontiguousarray(public_values, dtype=np.uint64))
            offs = (C.c_size_t * 13)()
            sizer = L.zkm_prove_segment_columns if self.by_columns else L.zkm_prove_segment
            _check(sizer(None, C.byref(cfg), self.ptr_arrays[-1], self.log_arrays[-1], self.pubs[-1].ctypes.data_as(u64p), self.pubs[-1].size, None,
                         offs, None, C.byref(err)), err)
            self.sizes.append(list(offs))
        self.proofs = [np.zeros(o[12], dtype=np.uint64) for o in self.sizes]
        self.chals = [np.zeros(2 * cfg.num_challenges, dtype=np.uint64) for _ in range(K)]
        self.tr = (C.c_void_p * K)(*[C.addressof(a) for a in self.ptr_arrays])
        self.lg = (C.c_void_p * K)(*[C.addressof(a) for a in self.log_arrays])
        self.pv = (C.c_void_p * K)(*[p.ctypes.data for p in self.pubs])
        self.npv = (C.c_size_t * K)(*[p.size for p in self.pubs])
        self.po = (C.c_void_p * K)(*[p.ctypes.data for p in self.proofs])
        self.co = (C.c_void_p * K)(*[p.ctypes.data for p in self.chals])

    def results(self):
        return [(self.proofs[i], self.chals[i], self.sizes[i]) for i in range(len(self.proofs))]


def pool_plan(nseg, workers, max_stack=0):
    """zkm_pool_plan: the sizes of the lock-step groups a pool call of nseg segments is cut into (a pure function of the library)."""
    L = load()
    n = L.zkm_pool_plan(nseg, workers, max_stack, None, 0)
    out = (C.c_size_t * max(1, n))()
    L.zkm_pool_plan(nseg, workers, max_stack, out, n)
    return [int(out[i]) for i in range(n)]


class Pool:
    """zkm_pool: `contexts_per_device` contexts on each of `devices`, one worker thread per context inside ONE process; a call cuts its
    segments into lock-step groups of at most max_stack and the workers pull them from a queue (include/zkm_hip.h).  The reference's
    one-process segment loop (prover/examples/utils/src/utils.rs:57-68, 105-133) over N GPUs."""

    def __init__(self, devices=(0,), contexts_per_device=1):
        self.L = load()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h, err = C.c_void_p(), C.c_char_p()
        _check(self.L.zkm_pool_create(devs, len(devices), contexts_per_device, C.byref(h), C.byref(err)), err)
        self.h = h

    def close(self):
        if self.h:
            self.L.zkm_pool_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def workers(self):
        return int(self.L.zkm_pool_workers(self.h))

    def device(self, worker):
        return int(self.L.zkm_pool_device(self.h, worker))

    def set_tuning(self, key, value):
        err = C.c_char_p()
        _check(self.L.zkm_pool_set_tuning(self.h, key.encode(), int(value), C.byref(err)), err)

    def standard_config(self):
        cfg = StarkConfig()
        self.L.zkm_standard_config(C.byref(cfg))
        return cfg

    def prove_segments(self, segments, max_stack=0, cfg=None):
        """zkm_pool_prove_segments[_columns]: segments as Context.prove_segments takes them (HOST arrays unless the pool has one device).
        Returns the list of (proofs, ctl_challenges, offsets), one per segment."""
        cfg = cfg or self.standard_config()
        m = _marshal_segments(self.L, segments, cfg)
        err = C.c_char_p()
        fn = self.L.zkm_pool_prove_segments_columns if m.by_columns else self.L.zkm_pool_prove_segments
        _check(fn(self.h, C.byref(cfg), len(segments), max_stack, m.tr, m.lg, m.pv, m.npv, m.po, m.co, C.byref(err)), err)
        return m.results()

    def last_assignment(self, segment):
        """(worker, group) that proved `segment` in the last call."""
        w, g = C.c_size_t(), C.c_size_t()
        if self.L.zkm_pool_last_assignment(self.h, segment, C.byref(w), C.byref(g)) != 0:
            raise ZkmError("zkm_pool_last_assignment: segment %d was not proven by the last call" % segment)
        return int(w.value), int(g.value)


class Context:
    """One GPU context (single-owner, like the reference's &mut Challenger / &mut TimingTree threading)."""

    def __init__(self, device=0):
        self.L = load()
        h = C.c_void_p()
        err = C.c_char_p()
        _check(self.L.zkm_ctx_create(device, C.byref(h), C.byref(err)), err)
        self.h = h

    def close(self):
        if self.h:
            for p in list(getattr(self, "_pinned", {}).values()):
                self.L.zkm_host_free(self.h, p)
            self._pinned = {}
            self.L.zkm_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def memory(self):
        """(live bytes, cached bytes) of the context's device allocator."""
        live, cached = C.c_size_t(), C.c_size_t()
        self.L.zkm_ctx_memory(self.h, C.byref(live), C.byref(cached))
        return live.value, cached.value

    def resident_bytes(self):
        """Of the live bytes: tables kept for reuse (twiddles, power tables; the commit lanes' included)."""
        return self.L.zkm_ctx_resident_bytes(self.h)

    def pinned_array(self, words):
        """A uint64 ndarray in page-locked host memory (zkm_host_alloc): uploads from it overlap with compute.  Freed with the
        context (or explicitly with free_pinned)."""
        p, err = C.c_void_p(), C.c_char_p()
        _check(self.L.zkm_host_alloc(self.h, words * 8, C.byref(p), C.byref(err)), err)
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(words,))
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def free_pinned(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None:
            self.L.zkm_host_free(self.h, p)

    def trim(self):
        """Return the allocator's cached blocks to the device."""
        self.L.zkm_ctx_trim(self.h)

    def set_tuning(self, key, value):
        """Kernel-selection thresholds (include/zkm_hip.h: zkm_ctx_set_tuning); every setting yields the same words."""
        err = C.c_char_p()
        _check(self.L.zkm_ctx_set_tuning(self.h, key.encode(), int(value), C.byref(err)), err)

    def synchronize(self):
        err = C.c_char_p()
        _check(self.L.zkm_ctx_synchronize(self.h, C.byref(err)), err)

    @property
    def stream(self):
        return self.L.zkm_ctx_stream(self.h)

    def alloc(self, words):
        return DeviceBuffer(self, words)

    # ---- primitives
    def ntt(self, cols, ncols, log_n, inverse=False, coset_shift=0):
        """In-place batched NTT (natural order in/out).  `cols`: host ndarray or device buffer."""
        err = C.c_char_p()
        _check(self.L.zkm_ntt(self.h, _data_ptr(cols), ncols, log_n, int(inverse), coset_shift, C.byref(err)), err)
        return cols

    def field_selftest(self, a, b):
        """(a + b, a - b, a 2^24, a 2^48, a 2^72, a b, a b [branch-free product]) mod p through the device's loose-arithmetic primitives,
        for arbitrary 64-bit words."""
        a, b = np.ascontiguousarray(a, dtype=np.uint64), np.ascontiguousarray(b, dtype=np.uint64)
        out = np.zeros(7 * a.size, dtype=np.uint64)
        err = C.c_char_p()
        _check(self.L.zkm_field_selftest(self.h, a.ctypes.data_as(u64p), b.ctypes.data_as(u64p), a.size, out.ctypes.data_as(u64p), C.byref(err)), err)
        return out.reshape(7, -1)

    def poseidon_permute_batch(self, states):
        k = (states.size if isinstance(states, np.ndarray) else states.words) // 12
        err = C.c_char_p()
        _check(self.L.zkm_poseidon_permute_batch(self.h, _data_ptr(states), k, C.byref(err)), err)
        return states

    def keccakf_batch(self, states):
        k = (states.size if isinstance(states, np.ndarray) else states.words) // 25
        err = C.c_char_p()
        _check(self.L.zkm_keccakf_batch(self.h, _data_ptr(states), k, C.byref(err)), err)
        return states

    def poseidon_trace(self, seed, num_perms, log_n, out=None):
        """PoseidonStark::generate_trace on the GPU; returns a DeviceBuffer of 262 x 2^log_n words."""
        out = out or self.alloc(POSEIDON_COLS << log_n)
        err = C.c_char_p()
        _check(self.L.zkm_poseidon_trace(self.h, seed, num_perms, log_n, _data_ptr(out), C.byref(err)), err)
        return out

    def keccak_sponge_trace(self, inputs, input_off, meta, log_n, out=None):
        """KeccakSpongeStark::generate_trace on the GPU (keccak_sponge_stark.rs:222-444).  inputs: uint8 array of all
        operations' bytes, input_off: nops+1 offsets, meta: nops x 4 (context, segment, virt_base, timestamp).
        Returns (DeviceBuffer of 470 x 2^log_n words, rows used)."""
        if isinstance(inputs, DeviceBuffer):      # message bytes already in HBM (packed into a word buffer): no upload inside the call
            in_ptr = C.c_void_p(inputs.ptr)
        else:
            inputs = np.ascontiguousarray(inputs, dtype=np.uint8)
            in_ptr = inputs.ctypes.data_as(C.c_void_p)
        input_off = np.ascontiguousarray(input_off, dtype=np.uint64)
        meta = np.ascontiguousarray(meta, dtype=np.uint64)
        nops = input_off.size - 1
        out = out or self.alloc(KECCAK_SPONGE_COLS << log_n)
        used = C.c_size_t()
        err = C.c_char_p()
        _check(self.L.zkm_keccak_sponge_trace(self.h, in_ptr, input_off.ctypes.data_as(u64p),
                                              meta.ctypes.data_as(u64p), nops, log_n, _data_ptr(out), C.byref(used), C.byref(err)), err)
        return out, used.value

    def poseidon_sponge_trace(self, inputs, input_off, meta, log_n, out=None):
        """PoseidonSpongeStark::generate_trace on the GPU (poseidon_sponge_stark.rs:186-381); arguments as keccak_sponge_trace.
        Returns (DeviceBuffer of 110 x 2^log_n words, rows used)."""
        inputs = np.ascontiguousarray(inputs, dtype=np.uint8)
        input_off = np.ascontiguousarray(input_off, dtype=np.uint64)
        meta = np.ascontiguousarray(meta, dtype=np.uint64)
        out = out or self.alloc(POSEIDON_SPONGE_COLS << log_n)
        used = C.c_size_t()
        err = C.c_char_p()
        _check(self.L.zkm_poseidon_sponge_trace(self.h, inputs.ctypes.data_as(C.c_void_p), input_off.ctypes.data_as(u64p),
                                                meta.ctypes.data_as(u64p), input_off.size - 1, log_n, _data_ptr(out), C.byref(used),
                                                C.byref(err)), err)
        return out, used.value

    def poseidon_trace_inputs(self, inputs, timestamps, log_n, out=None):
        """PoseidonStark::generate_trace for explicit inputs (num_perms x 12) and timestamps; DeviceBuffer of 262 x 2^log_n."""
        inputs = np.ascontiguousarray(inputs, dtype=np.uint64).reshape(-1, 12)
        timestamps = np.ascontiguousarray(timestamps, dtype=np.uint64)
        out = out or self.alloc(POSEIDON_COLS << log_n)
        err = C.c_char_p()
        _check(self.L.zkm_poseidon_trace_inputs(self.h, _data_ptr(inputs), _data_ptr(timestamps), len(inputs), log_n, _data_ptr(out),
                                                C.byref(err)), err)
        return out

    def sha_extend_trace(self, inputs, timestamps, log_n, out=None):
        """ShaExtendStark::generate_trace on the GPU: inputs nrows x 16 bytes, one timestamp per row; 78 x 2^log_n words."""
        inputs = np.ascontiguousarray(inputs, dtype=np.uint8).reshape(-1, 16)
        timestamps = np.ascontiguousarray(timestamps, dtype=np.uint64)
        out = out or self.alloc(78 << log_n)
        err = C.c_char_p()
        _check(self.L.zkm_sha_extend_trace(self.h, inputs.ctypes.data_as(C.c_void_p), _data_ptr(timestamps), len(inputs), log_n,
                                           _data_ptr(out), C.byref(err)), err)
        return out

    def sha_extend_sponge_trace(self, w16, meta, log_n, out=None):
        """ShaExtendSpongeStark::generate_trace on the GPU for complete schedules: w16 nblocks x 16 uint32, meta nblocks x 4."""
        w16 = np.ascontiguousarray(w16, dtype=np.uint32).reshape(-1, 16)
        meta = np.ascontiguousarray(meta, dtype=np.uint64).reshape(-1, 4)
        out = out or self.alloc(76 << log_n)
        err = C.c_char_p()
        _check(self.L.zkm_sha_extend_sponge_trace(self.h, w16.ctypes.data_as(C.c_void_p), _data_ptr(meta), len(w16), log_n,
                                                  _data_ptr(out), C.byref(err)), err)
        return out

    def _sha_compress(self, fn, cols, hx, w, meta, log_n, out):
        hx = np.ascontiguousarray(hx, dtype=np.uint32).reshape(-1, 8)
        w = np.ascontiguousarray(w, dtype=np.uint32).reshape(-1, 64)
        meta = np.ascontiguousarray(meta, dtype=np.uint64).reshape(-1, 8)
        out = out or self.alloc(cols << log_n)
        err = C.c_char_p()
        _check(fn(self.h, hx.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), _data_ptr(meta), len(hx), log_n, _data_ptr(out),
                  C.byref(err)), err)
        return out

    def sha_compress_trace(self, hx, w, meta, log_n, out=None):
        """ShaCompressStark::generate_trace on the GPU: 65 rows per compression, 224 x 2^log_n words."""
        return self._sha_compress(self.L.zkm_sha_compress_trace, 224, hx, w, meta, log_n, out)

    def sha_compress_sponge_trace(self, hx, w, meta, log_n, out=None):
        """ShaCompressSpongeStark::generate_trace on the GPU: one row per compression, 127 x 2^log_n words."""
        return self._sha_compress(self.L.zkm_sha_compress_sponge_trace, 127, hx, w, meta, log_n, out)

    def keccak_trace(self, inputs, timestamps, log_n, out=None):
        """KeccakStark::generate_trace on the GPU (keccak/keccak_stark.rs:62-236).  inputs: nperms x 25 uint64, timestamps:
        nperms uint64 (ndarrays or DeviceBuffers).  Returns a DeviceBuffer of 2431 x 2^log_n words."""
        if isinstance(inputs, DeviceBuffer):
            nperms = inputs.words // 25
        else:
            inputs = np.ascontiguousarray(inputs, dtype=np.uint64).reshape(-1, 25)
            nperms = len(inputs)
        if not isinstance(timestamps, DeviceBuffer):
            timestamps = np.ascontiguousarray(timestamps, dtype=np.uint64)
        out = out or self.alloc(KECCAK_COLS << log_n)
        err = C.c_char_p()
        _check(self.L.zkm_keccak_trace(self.h, _data_ptr(inputs), _data_ptr(timestamps), nperms, log_n, _data_ptr(out), C.byref(err)), err)
        return out

    def logic_trace(self, ops, log_n, out=None):
        """LogicStark::generate_trace on the GPU (logic.rs:150-183).  ops: nops x 3 uint32 (op, input0, input1).
        Returns a DeviceBuffer of 69 x 2^log_n words."""
        ops = np.ascontiguousarray(ops, dtype=np.uint32).reshape(-1, 3)
        out = out or self.alloc(LOGIC_COLS << log_n)
        err = C.c_char_p()
        _check(self.L.zkm_logic_trace(self.h, ops.ctypes.data_as(C.c_void_p), len(ops), log_n, _data_ptr(out), C.byref(err)), err)
        return out

    # ---- profiling
    def profile(self, on=True):
        self.L.zkm_profile_enable(self.h, int(on))

    def profile_reset(self):
        self.L.zkm_profile_reset(self.h)

    def profile_records(self):
        out = {}
        for i in range(self.L.zkm_profile_count(self.h)):
            name, n, ms = C.c_char_p(), C.c_uint64(), C.c_double()
            self.L.zkm_profile_get(self.h, i, C.byref(name), C.byref(n), C.byref(ms))
            out[name.value.decode()] = (n.value, ms.value)
        return out

    # ---- STARK
    def standard_config(self):
        cfg = StarkConfig()
        self.L.zkm_standard_config(C.byref(cfg))
        return cfg

    def proof_words(self, cfg, log_n, ncols, naux, nctl):
        return self.L.zkm_proof_words(C.byref(cfg), log_n, ncols, naux, nctl)

    def stage_trace(self, values, ncols, log_n, canonical=True):
        """zkm_trace_stage[_columns]: queue the upload of a host matrix (one array, or a list of per-column arrays) on the copy streams and
        return at once; the StagedTrace goes where a prove call of this context takes a trace.  The host arrays must stay alive and
        unchanged until ready() or free()."""
        h, err = C.c_void_p(), C.c_char_p()
        if isinstance(values, (list, tuple)):
            keep = [np.ascontiguousarray(c, dtype=np.uint64) for c in values]
            cols = (C.c_void_p * len(keep))(*[c.ctypes.data for c in keep])
            _check(self.L.zkm_trace_stage_columns(self.h, cols, ncols, log_n, 1 if canonical else 0, C.byref(h), C.byref(err)), err)
        else:
            keep = np.ascontiguousarray(values, dtype=np.uint64)
            assert keep.size == ncols << log_n
            _check(self.L.zkm_trace_stage(self.h, _np_ptr(keep), ncols, log_n, 1 if canonical else 0, C.byref(h), C.byref(err)), err)
        st = StagedTrace(self, h, ncols << log_n)
        st._keep = keep
        return st

    def stage_segment(self, traces, log_ns, canonical=True):
        """zkm_segment_stage[_columns]: all twelve tables of one segment (host arrays, or lists of per-column arrays) in ONE call.  Returns a
        StagedTrace whose .tables() are the twelve device pointers to hand to prove_segments as that segment's traces."""
        assert len(traces) == 12 and len(log_ns) == 12
        h, err = C.c_void_p(), C.c_char_p()
        lg = (C.c_uint * 12)(*[int(x) for x in log_ns])
        if all(isinstance(t, (list, tuple)) for t in traces):
            keep = [[np.ascontiguousarray(c, dtype=np.uint64) for c in t] for t in traces]
            cols = [(C.c_void_p * len(t))(*[c.ctypes.data for c in t]) for t in keep]
            tabs = (C.c_void_p * 12)(*[C.addressof(c) for c in cols])
            keep = (keep, cols)
            _check(self.L.zkm_segment_stage_columns(self.h, tabs, lg, 1 if canonical else 0, C.byref(h), C.byref(err)), err)
        else:
            keep = [np.ascontiguousarray(t, dtype=np.uint64) for t in traces]
            tabs = (C.c_void_p * 12)(*[t.ctypes.data for t in keep])
            _check(self.L.zkm_segment_stage(self.h, tabs, lg, 1 if canonical else 0, C.byref(h), C.byref(err)), err)
        st = StagedTrace(self, h, 0)
        st._keep = keep
        return st

    def prove_single_table(self, trace, log_n, aux, num_helpers, challenger=None, cfg=None, ncols=POSEIDON_COLS,
                           trace_batch=None, naux=None, table_id=TABLE_POSEIDON):
        """prove_single_table (prover.rs:441-641).  Returns the flat proof (include/zkm_hip.h layout)."""
        cfg = cfg or self.standard_config()
        ch = challenger if challenger is not None else Challenger()
        if naux is None:
            naux = (aux.size if isinstance(aux, np.ndarray) else aux.words) >> log_n
        nh = (C.c_uint32 * len(num_helpers))(*num_helpers)
        proof = np.zeros(self.proof_words(cfg, log_n, ncols, naux, len(num_helpers)), dtype=np.uint64)
        err = C.c_char_p()
        _check(self.L.zkm_prove_single_table(self.h, table_id, C.byref(cfg), _data_ptr(trace) if trace is not None else None,
                                             ncols, log_n, trace_batch.h if trace_batch is not None else None, _data_ptr(aux),
                                             naux, nh, len(num_helpers), C.byref(ch), proof.ctypes.data_as(u64p),
                                             C.byref(err)), err)
        return proof

    def prove_single_tables(self, traces, log_n, auxs, num_helpers, cfg=None, ncols=POSEIDON_COLS, table_id=TABLE_POSEIDON, challengers=None):
        """zkm_prove_single_tables: K proofs of the same table in lock-step.  traces[k] / auxs[k] as prove_single_table takes them
        (auxs may be ONE array used for every proof).  Returns the list of K proof blobs."""
        cfg = cfg or self.standard_config()
        K = len(traces)
        if not isinstance(auxs, (list, tuple)):
            auxs = [auxs] * K
        chs = challengers if challengers is not None else [Challenger() for _ in range(K)]
        naux = (auxs[0].size if isinstance(auxs[0], np.ndarray) else auxs[0].words) >> log_n
        nh = (C.c_uint32 * len(num_helpers))(*num_helpers)
        words = self.proof_words(cfg, log_n, ncols, naux, len(num_helpers))
        proofs = [np.zeros(words, dtype=np.uint64) for _ in range(K)]
        tp = (C.c_void_p * K)(*[_data_ptr(t).value for t in traces])
        ap = (C.c_void_p * K)(*[_data_ptr(a).value for a in auxs])
        cp_ = (C.c_void_p * K)(*[C.addressof(ch) for ch in chs])
        pp = (C.c_void_p * K)(*[p.ctypes.data for p in proofs])
        err = C.c_char_p()
        _check(self.L.zkm_prove_single_tables(self.h, table_id, C.byref(cfg), K, tp, ncols, log_n, ap, naux, nh, len(num_helpers), cp_, pp,
                                              C.byref(err)), err)
        return proofs

    # ---- cross-table lookups (descriptor builders: zkm_amd/ctl.py)
    def ctl_data(self, ctl_table, zs, colset_ids, trace, ncols, log_n, out=None):
        """cross_table_lookup_data for one table (cross_table_lookup.rs:634-872): helper columns ++ Z columns."""
        naux = int(zs["num_helpers"].sum()) + len(zs)
        host_out = out is None
        if host_out:
            out = np.zeros(naux << log_n, dtype=np.uint64)
        st = ctl_table.pack()
        err = C.c_char_p()
        _check(self.L.zkm_ctl_data(self.h, C.addressof(st), zs.ctypes.data, colset_ids.ctypes.data, len(zs), _data_ptr(trace), ncols,
                                   log_n, _data_ptr(out), C.byref(err)), err)
        return out

    def lookup_helper_columns(self, ctl_table, colset_ids, table_col, freq_col, challenge, trace, ncols, log_n):
        """lookup_helper_columns (lookup.rs:46-124) for one Lookup and one challenge: helper columns then Z."""
        ids = np.ascontiguousarray(colset_ids, dtype=np.uint32)
        out = np.zeros(((len(ids) + 1) // 2 + 1) << log_n, dtype=np.uint64)
        st = ctl_table.pack()
        err = C.c_char_p()
        _check(self.L.zkm_lookup_helper_columns(self.h, C.addressof(st), ids.ctypes.data, len(ids), table_col, freq_col, challenge,
                                                _data_ptr(trace), ncols, log_n, _np_ptr(out), C.byref(err)), err)
        return out

    def prove_single_table_ctl(self, trace, log_n, aux, ctl_table, zs, colset_ids, challenger=None, cfg=None, ncols=POSEIDON_COLS,
                               trace_batch=None, table_id=TABLE_POSEIDON, lookup_challenges=None):
        """lookup_challenges: the CTL betas, required for tables with their own lookups (Memory)."""
        cfg = cfg or self.standard_config()
        ch = challenger if challenger is not None else Challenger()
        naux = (aux.size if isinstance(aux, np.ndarray) else aux.words) >> log_n
        nl = self.L.zkm_num_lookup_columns(table_id, C.byref(cfg))
        proof = np.zeros(self.proof_words(cfg, log_n, ncols, naux + nl, len(zs)), dtype=np.uint64)
        st = ctl_table.pack()
        err = C.c_char_p()
        lk = None if lookup_challenges is None else np.ascontiguousarray(lookup_challenges, dtype=np.uint64)
        _check(self.L.zkm_prove_single_table_ctl(self.h, table_id, C.byref(cfg), _data_ptr(trace) if trace is not None else None, ncols,
                                                 log_n, trace_batch.h if trace_batch is not None else None, _data_ptr(aux), naux,
                                                 C.addressof(st), zs.ctypes.data, colset_ids.ctypes.data, len(zs),
                                                 None if lk is None else lk.ctypes.data_as(u64p), C.byref(ch),
                                                 proof.ctypes.data_as(u64p), C.byref(err)), err)
        return proof

    def check_constraints(self, trace, log_n, aux, ctl_table, zs, colset_ids, alphas, cfg=None, ncols=POSEIDON_COLS, table_id=TABLE_POSEIDON,
                          lookup_challenges=None):
        """check_constraints (prover.rs:793-910): the table's whole vanishing polynomial on every row of the trace domain.  aux = all
        auxiliary columns (lookup helper columns ++ CTL helper columns ++ Zs).  Returns None when every constraint holds, else the
        first failing row."""
        cfg = cfg or self.standard_config()
        naux = (aux.size if isinstance(aux, np.ndarray) else aux.words) >> log_n
        al = np.ascontiguousarray(alphas, dtype=np.uint64)
        lk = None if lookup_challenges is None else np.ascontiguousarray(lookup_challenges, dtype=np.uint64)
        first = C.c_uint64(0)
        err = C.c_char_p()
        if ctl_table is not None:
            st = ctl_table.pack()
            tp, zp, ip, nz = C.addressof(st), zs.ctypes.data, colset_ids.ctypes.data, len(zs)
        else:     # fake CTL shape: zs = num_helpers list
            from . import ctl as zc
            z = np.zeros(len(zs), dtype=zc.CTLZ_DT)
            z["num_helpers"] = zs
            tp, zp, ip, nz = None, z.ctypes.data, None, len(zs)
        rc = self.L.zkm_check_constraints(self.h, table_id, C.byref(cfg), _data_ptr(trace), ncols, log_n, _data_ptr(aux), naux, tp, zp, ip, nz,
                                          None if lk is None else lk.ctypes.data_as(u64p), al.ctypes.data_as(u64p), al.size, C.byref(first),
                                          C.byref(err))
        if rc == 0:
            return None
        msg = err.value.decode() if err.value else ""
        if "Constraint failed" not in msg:
            _check(rc, err)
        return int(first.value)

    def prove_with_traces(self, tables, ctls, public_values=(), cfg=None):
        """prove_with_traces (prover.rs:130-232).  tables: list of (table_id, trace (ndarray | DeviceBuffer), ncols, log_n, CtlTable);
        ctls: list of (looking=[(table, colset)..], looked=(table, colset)).  Returns (proofs, ctl_challenges, offsets)."""
        from . import ctl as zc
        cfg = cfg or self.standard_config()
        # (a trace given as a LIST of per-column arrays goes through zkm_table_input.columns: one pointer per column)
        packed = [(tid, [_data_ptr(col).value for col in tr] if isinstance(tr, (list, tuple)) else _data_ptr(tr).value, ncols, log_n, ct)
                  for (tid, tr, ncols, log_n, ct) in tables]
        tarr, keep = zc.pack_tables(packed)
        carr, sides = zc.pack_ctls(ctls)
        offs = (C.c_size_t * (len(tables) + 1))()
        total = self.L.zkm_all_proof_words(C.byref(cfg), tarr, len(tables), carr.ctypes.data, sides.ctypes.data, len(carr), offs)
        if total == 0 and len(tables):
            raise ZkmError("malformed cross-table lookups")
        proofs = np.zeros(total, dtype=np.uint64)
        chal = np.zeros(2 * cfg.num_challenges, dtype=np.uint64)
        pub = np.ascontiguousarray(public_values, dtype=np.uint64)
        err = C.c_char_p()
        _check(self.L.zkm_prove_with_traces(self.h, C.byref(cfg), tarr, len(tables), carr.ctypes.data, sides.ctypes.data, len(carr),
                                            pub.ctypes.data_as(u64p), pub.size, proofs.ctypes.data_as(u64p), chal.ctypes.data_as(u64p),
                                            C.byref(err)), err)
        return proofs, chal, list(offs)

    def prove_segment(self, traces, log_ns, public_values=(), cfg=None):
        """prove_with_traces (prover.rs:130-232) on the twelve tables of Table::all() (all_stark.rs:117-134) with the AllStark
        description that ships inside the library: traces[t] = ndarray or DeviceBuffer of table t in enum order, log_ns[t] its
        height.  Returns (proofs, ctl_challenges, offsets)."""
        cfg = cfg or self.standard_config()
        assert len(traces) == 12 and len(log_ns) == 12
        if all(isinstance(t, (list, tuple)) for t in traces):
            return self._prove_segment_columns(traces, log_ns, public_values, cfg)
        keep = [t if isinstance(t, (DeviceBuffer, StagedTrace)) else np.ascontiguousarray(t, dtype=np.uint64) for t in traces]
        ptrs = (C.c_void_p * 12)(*[_data_ptr(t).value for t in keep])
        lg = (C.c_uint * 12)(*[int(x) for x in log_ns])
        pub = np.ascontiguousarray(public_values, dtype=np.uint64)
        offs = (C.c_size_t * 13)()
        err = C.c_char_p()
        _check(self.L.zkm_prove_segment(None, C.byref(cfg), ptrs, lg, pub.ctypes.data_as(u64p), pub.size, None, offs, None, C.byref(err)), err)
        proofs = np.zeros(offs[12], dtype=np.uint64)
        chal = np.zeros(2 * cfg.num_challenges, dtype=np.uint64)
        _check(self.L.zkm_prove_segment(self.h, C.byref(cfg), ptrs, lg, pub.ctypes.data_as(u64p), pub.size, proofs.ctypes.data_as(u64p), offs,
                                        chal.ctypes.data_as(u64p), C.byref(err)), err)
        return proofs, chal, list(offs)

    def prove_segments(self, segments, cfg=None):
        """zkm_prove_segments: K independent segments in lock-step (one launch per stage for all segments whose table has the same
        height).  segments = list of (traces, log_ns, public_values) as prove_segment takes them.  Returns a list of
        (proofs, ctl_challenges, offsets), one per segment -- each equal to prove_segment's result for that segment alone."""
        cfg = cfg or self.standard_config()
        m = _marshal_segments(self.L, segments, cfg)
        err = C.c_char_p()
        fn = self.L.zkm_prove_segments_columns if m.by_columns else self.L.zkm_prove_segments
        _check(fn(self.h, C.byref(cfg), len(segments), m.tr, m.lg, m.pv, m.npv, m.po, m.co, C.byref(err)), err)
        return m.results()

    def _prove_segment_columns(self, traces, log_ns, public_values, cfg):
        """zkm_prove_segment_columns: traces[t] = list of per-column arrays (each its own allocation, like Vec<PolynomialValues<F>>)."""
        keep = [[c if isinstance(c, (DeviceBuffer, StagedTrace)) else np.ascontiguousarray(c, dtype=np.uint64) for c in t] for t in traces]
        cols = [(C.c_void_p * len(t))(*[_data_ptr(c).value for c in t]) for t in keep]
        tabs = (C.c_void_p * 12)(*[C.addressof(c) for c in cols])
        lg = (C.c_uint * 12)(*[int(x) for x in log_ns])
        pub = np.ascontiguousarray(public_values, dtype=np.uint64)
        offs = (C.c_size_t * 13)()
        err = C.c_char_p()
        _check(self.L.zkm_prove_segment_columns(None, C.byref(cfg), tabs, lg, pub.ctypes.data_as(u64p), pub.size, None, offs, None, C.byref(err)), err)
        proofs = np.zeros(offs[12], dtype=np.uint64)
        chal = np.zeros(2 * cfg.num_challenges, dtype=np.uint64)
        _check(self.L.zkm_prove_segment_columns(self.h, C.byref(cfg), tabs, lg, pub.ctypes.data_as(u64p), pub.size, proofs.ctypes.data_as(u64p),
                                                offs, chal.ctypes.data_as(u64p), C.byref(err)), err)
        return proofs, chal, list(offs)

    def prove_segment_image(self, image, cfg=None):
        """Prove every table of a ZKMTRACE segment image (see segment_image()).  Returns (proofs, ctl_challenges, offsets)."""
        cfg = cfg or self.standard_config()
        image = np.ascontiguousarray(image, dtype=np.uint64)
        ntables = int(image[2]) if image.size > 2 else 0
        offs = (C.c_size_t * (ntables + 1))()
        total = C.c_size_t()
        err = C.c_char_p()
        _check(self.L.zkm_prove_segment_image(self.h, C.byref(cfg), image.ctypes.data_as(u64p), image.size, None, C.byref(total), offs,
                                              None, C.byref(err)), err)
        proofs = np.zeros(total.value, dtype=np.uint64)
        chal = np.zeros(2 * cfg.num_challenges, dtype=np.uint64)
        _check(self.L.zkm_prove_segment_image(self.h, C.byref(cfg), image.ctypes.data_as(u64p), image.size, proofs.ctypes.data_as(u64p),
                                              C.byref(total), offs, chal.ctypes.data_as(u64p), C.byref(err)), err)
        return proofs, chal, list(offs)

    def fri_prove(self, oracles, batches, challenger, cfg=None):
        """PolynomialBatch::prove_openings for an arbitrary FriInstanceInfo: oracles = list of PolynomialBatch, batches = list of
        ((point_c0, point_c1), [(oracle_index, polynomial_index), ...]).  Returns the FRI proof blob (layout in zkm_hip.h)."""
        cfg = cfg or self.standard_config()
        orc = (C.c_void_p * len(oracles))(*[o.h.value for o in oracles])
        cols = (C.c_size_t * len(oracles))(*[o.ncols for o in oracles])
        words = self.L.zkm_fri_proof_words(C.byref(cfg), oracles[0].log_n, cols, len(oracles))
        if not words:
            raise ZkmError("zkm_fri_proof_words: unsupported configuration")

        keep, arr = [], (FriBatch * len(batches))()
        for i, (pt, polys) in enumerate(batches):
            a = np.array(polys, dtype=np.uint32).reshape(-1, 2)
            keep.append(a)
            arr[i] = FriBatch((C.c_uint64 * 2)(int(pt[0]), int(pt[1])), a.ctypes.data, len(a))
        out = np.zeros(words, dtype=np.uint64)
        err = C.c_char_p()
        _check(self.L.zkm_fri_prove(self.h, C.byref(cfg), C.cast(orc, C.POINTER(C.c_void_p)), len(oracles), C.cast(arr, C.c_void_p), len(batches),
                                    C.byref(challenger), out.ctypes.data_as(u64p), C.byref(err)), err)
        return out

    def prove_openings(self, trace_batch, aux_batch, quot_batch, nctl_zs, challenger=None, cfg=None):
        """PolynomialBatch::prove_openings for the STARK FRI instance on three existing commitments (BASELINE config 4)."""
        cfg = cfg or self.standard_config()
        ch = challenger if challenger is not None else Challenger()
        proof = np.zeros(self.proof_words(cfg, trace_batch.log_n, trace_batch.ncols, aux_batch.ncols, nctl_zs), dtype=np.uint64)
        err = C.c_char_p()
        _check(self.L.zkm_prove_openings(self.h, C.byref(cfg), trace_batch.h, aux_batch.h, quot_batch.h, nctl_zs, C.byref(ch),
                                         proof.ctypes.data_as(u64p), C.byref(err)), err)
        return proof

    def quotient(self, trace_batch, aux_batch, num_helpers, alphas, table_id=TABLE_POSEIDON):
        nh = (C.c_uint32 * len(num_helpers))(*num_helpers)
        al = np.ascontiguousarray(alphas, dtype=np.uint64)
        out = np.zeros(al.size * (2 << trace_batch.log_n), dtype=np.uint64)
        err = C.c_char_p()
        _check(self.L.zkm_quotient(self.h, table_id, trace_batch.h, aux_batch.h, nh, len(num_helpers),
                                   al.ctypes.data_as(u64p), al.size, _np_ptr(out), C.byref(err)), err)
        return out

    def eval_openings(self, batch, zeta):
        z = np.array(zeta, dtype=np.uint64)
        out = np.zeros(2 * batch.ncols, dtype=np.uint64)
        err = C.c_char_p()
        _check(self.L.zkm_eval_openings(self.h, batch.h, z.ctypes.data_as(u64p), out.ctypes.data_as(u64p), C.byref(err)), err)
        return out


class PolynomialBatch:
    """== plonky2 PolynomialBatch<F, PoseidonGoldilocksConfig, 2> living in HBM."""

    def __init__(self, ctx, h, ncols, log_n, rate_bits, cap_height):
        self.ctx, self.h, self.ncols, self.log_n, self.rate_bits, self.cap_height = ctx, h, ncols, log_n, rate_bits, cap_height

    @classmethod
    def from_values(cls, ctx, values, ncols, log_n, rate_bits=2, cap_height=4):
        h = C.c_void_p()
        err = C.c_char_p()
        _check(ctx.L.zkm_batch_commit_values(ctx.h, _data_ptr(values), ncols, log_n, rate_bits, cap_height, C.byref(h),
                                             C.byref(err)), err)
        return cls(ctx, h, ncols, log_n, rate_bits, cap_height)

    @classmethod
    def from_columns(cls, ctx, columns, log_n, values=True, rate_bits=2, cap_height=4):
        """from_values / from_coeffs from one array per column (zkm_batch_commit_columns): the shape of the reference's
        Vec<PolynomialValues<F>> -- nothing is flattened on the host."""
        keep = [c if isinstance(c, (DeviceBuffer, StagedTrace)) else np.ascontiguousarray(c, dtype=np.uint64) for c in columns]
        ptrs = (C.c_void_p * len(keep))(*[_data_ptr(c).value for c in keep])
        h = C.c_void_p()
        err = C.c_char_p()
        _check(ctx.L.zkm_batch_commit_columns(ctx.h, ptrs, len(keep), log_n, int(bool(values)), rate_bits, cap_height, C.byref(h),
                                              C.byref(err)), err)
        return cls(ctx, h, len(keep), log_n, rate_bits, cap_height)

    @classmethod
    def from_coeffs(cls, ctx, coeffs, ncols, log_n, rate_bits=2, cap_height=4):
        h = C.c_void_p()
        err = C.c_char_p()
        _check(ctx.L.zkm_batch_commit_coeffs(ctx.h, _data_ptr(coeffs), ncols, log_n, rate_bits, cap_height, C.byref(h),
                                             C.byref(err)), err)
        return cls(ctx, h, ncols, log_n, rate_bits, cap_height)

    def free(self):
        if self.h:
            self.ctx.L.zkm_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            if self.ctx.h:
                self.free()
        except Exception:
            pass

    @property
    def lde_bits(self):
        return self.log_n + self.rate_bits

    def cap(self):
        out = np.zeros(4 << self.cap_height, dtype=np.uint64)
        assert self.ctx.L.zkm_batch_cap(self.h, out.ctypes.data_as(u64p)) == 0
        return out

    def coeffs(self):
        out = np.zeros(self.ncols << self.log_n, dtype=np.uint64)
        assert self.ctx.L.zkm_batch_coeffs(self.h, _np_ptr(out)) == 0
        return out

    def lde_row(self, natural_index):
        out = np.zeros(self.ncols, dtype=np.uint64)
        assert self.ctx.L.zkm_batch_lde_row(self.h, natural_index, out.ctypes.data_as(u64p)) == 0
        return out

    def lde_rows(self, index_start, step, count):
        """get_lde_values_packed(index_start, step) for `count` consecutive indices: count x ncols."""
        out = np.zeros(count * self.ncols, dtype=np.uint64)
        if self.ctx.L.zkm_batch_lde_rows(self.h, index_start, step, count, _np_ptr(out)) != 0:
            raise ZkmError("zkm_batch_lde_rows failed")
        return out.reshape(count, self.ncols)

    def leaf(self, i):
        out = np.zeros(self.ncols, dtype=np.uint64)
        assert self.ctx.L.zkm_batch_leaf(self.h, i, out.ctypes.data_as(u64p)) == 0
        return out

    def merkle_path(self, i):
        out = np.zeros(4 * (self.lde_bits - self.cap_height), dtype=np.uint64)
        assert self.ctx.L.zkm_batch_merkle_path(self.h, i, out.ctypes.data_as(u64p)) == 0
        return out

    def digest_layer(self, level):
        out = np.zeros(4 << (self.lde_bits - level), dtype=np.uint64)
        assert self.ctx.L.zkm_batch_digest_layer(self.h, level, out.ctypes.data_as(u64p)) == 0
        return out


def challenger_new():
    ch = Challenger()
    load().zkm_challenger_init(C.byref(ch))
    return ch


def challenger_observe(ch, elems):
    a = np.ascontiguousarray(elems, dtype=np.uint64)
    load().zkm_challenger_observe(C.byref(ch), a.ctypes.data_as(u64p), a.size)


def challenger_get(ch):
    return load().zkm_challenger_get(C.byref(ch))


def proof_layout(proof):
    """Field offsets of a proof blob (zkm_proof_get_layout / zkm_proof_get_query_layout).  No GPU needed."""
    L = load()
    proof = np.ascontiguousarray(proof, dtype=np.uint64)
    lay, q = ProofLayout(), ProofQueryLayout()
    if L.zkm_proof_get_layout(proof.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(lay)) or \
            L.zkm_proof_get_query_layout(proof.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(q)):
        raise ZkmError("not a proof blob")
    return lay, q


def segment_image(tables, ctls, public_values=()):
    """Serialise a segment (host traces + cross-table lookups, same arguments as Context.prove_with_traces) into one ZKMTRACE image
    (zkm_segment_image_write).  No GPU needed."""
    from . import ctl as zc
    L = load()
    packed = [(tid, _data_ptr(tr).value, ncols, log_n, ct) for (tid, tr, ncols, log_n, ct) in tables]
    tarr, keep = zc.pack_tables(packed)
    carr, sides = zc.pack_ctls(ctls)
    pub = np.ascontiguousarray(public_values, dtype=np.uint64)
    words = L.zkm_segment_image_words(tarr, len(tables), carr.ctypes.data, sides.ctypes.data, len(carr), pub.size)
    img = np.zeros(words, dtype=np.uint64)
    err = C.c_char_p()
    u64p = C.POINTER(C.c_uint64)
    _check(L.zkm_segment_image_write(tarr, len(tables), carr.ctypes.data, sides.ctypes.data, len(carr), pub.ctypes.data_as(u64p), pub.size,
                                     img.ctypes.data_as(u64p), C.byref(err)), err)
    return img
