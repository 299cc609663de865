"""ctypes binding of the CPU oracle (oracle/libzkm_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as
the checker.  Nothing under zkm_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libzkm_oracle.so")

P = 0xFFFFFFFF00000001
POSEIDON_COLS = 262
u64p = C.POINTER(C.c_uint64)


def build(force=False):
    if force or not os.path.exists(_LIB) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB)
            for f in os.listdir(_HERE) if f.endswith((".c", ".h", ".inc"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
    return _LIB


class Challenger(C.Structure):
    _fields_ = [("state", C.c_uint64 * 12), ("in_buf", C.c_uint64 * 8), ("out_buf", C.c_uint64 * 8),
                ("n_in", C.c_uint32), ("n_out", C.c_uint32)]


class StarkConfig(C.Structure):
    _fields_ = [(n, C.c_uint) for n in
                ("rate_bits", "cap_height", "pow_bits", "num_challenges", "num_queries", "arity_bits", "final_poly_bits")]


def _ptr(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def _default_threads():
    """OpenMP threads for the oracle.  The parallel regions are short (per column / per row loops of small tables); on a 256-core
    host the fork/join cost of 256 threads per region dominates (a 12-table test segment: 157 s with 256 threads, 25 s with 8),
    so the default is capped.  OMP_NUM_THREADS in the environment wins; Oracle.set_threads() changes it at run time."""
    return min(os.cpu_count() or 1, 32)


class Oracle:
    def __init__(self, threads=None):
        if "OMP_NUM_THREADS" not in os.environ and threads is None:
            threads = _default_threads()
        self.lib = C.CDLL(build())
        self._gomp = None
        if threads is not None:
            self.set_threads(threads)
        L = self.lib
        L.zko_gl_mul.restype = C.c_uint64
        L.zko_gl_mul.argtypes = [C.c_uint64, C.c_uint64]
        L.zko_gl_inv.restype = C.c_uint64
        L.zko_gl_inv.argtypes = [C.c_uint64]
        L.zko_gl_pow.restype = C.c_uint64
        L.zko_gl_pow.argtypes = [C.c_uint64, C.c_uint64]
        L.zko_gl_root_of_unity.restype = C.c_uint64
        L.zko_gl_root_of_unity.argtypes = [C.c_uint]
        L.zko_ntt.argtypes = [u64p, C.c_size_t, C.c_uint, C.c_int, C.c_uint64]
        L.zko_batch_from_values.restype = C.c_void_p
        L.zko_batch_from_values.argtypes = [u64p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint]
        L.zko_batch_from_coeffs.restype = C.c_void_p
        L.zko_batch_from_coeffs.argtypes = [u64p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint]
        L.zko_batch_free.argtypes = [C.c_void_p]
        for f in ("zko_batch_cap", "zko_batch_coeffs"):
            getattr(L, f).argtypes = [C.c_void_p, u64p]
        for f in ("zko_batch_lde_row", "zko_batch_leaf", "zko_batch_merkle_path"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_size_t, u64p]
        L.zko_batch_digest_layer.argtypes = [C.c_void_p, C.c_uint, u64p]
        L.zko_poseidon_trace.argtypes = [C.c_uint64, C.c_size_t, C.c_uint, u64p]
        L.zko_proof_words.restype = C.c_size_t
        L.zko_proof_words.argtypes = [C.POINTER(StarkConfig), C.c_uint, C.c_size_t, C.c_size_t, C.c_size_t]
        L.zko_prove_single_table.restype = C.c_int
        L.zko_prove_single_table.argtypes = [C.c_int, C.POINTER(StarkConfig), u64p, C.c_size_t, C.c_uint, u64p, C.c_size_t,
                                             C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(Challenger), u64p,
                                             C.POINTER(C.c_double)]
        L.zko_verify_single_table.restype = C.c_int
        L.zko_verify_single_table.argtypes = [C.c_int, C.POINTER(StarkConfig), u64p, C.c_size_t, C.c_size_t,
                                              C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(Challenger)]
        L.zko_quotient_poseidon.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_size_t, u64p, C.c_size_t, u64p]
        L.zko_poseidon_eval_row.argtypes = [u64p, u64p, C.c_size_t, u64p]
        L.zko_challenger_get.restype = C.c_uint64

    def set_threads(self, n):
        """omp_set_num_threads for the oracle's parallel regions (libgomp is already loaded by the oracle library)."""
        if self._gomp is None:
            self._gomp = C.CDLL("libgomp.so.1")
        self._gomp.omp_set_num_threads(int(n))
        self.threads = int(n)

    def set_wide_threads(self, n):
        """Threads of the long row-/column-parallel regions of a big commitment (leaf hashing, per-column LDE, lower Merkle levels):
        they scale to every host core, the short regions do not.  0: same as set_threads."""
        self.lib.zko_set_wide_threads(int(n))
        self.wide_threads = int(n)

    def get_threads(self):
        if self._gomp is None:
            self._gomp = C.CDLL("libgomp.so.1")
        return int(self._gomp.omp_get_max_threads())

    # ---- primitives
    def gl_mul(self, a, b):
        return self.lib.zko_gl_mul(a, b)

    def gl_inv(self, a):
        return self.lib.zko_gl_inv(a)

    def gl_pow(self, a, e):
        return self.lib.zko_gl_pow(a, e)

    def root_of_unity(self, k):
        return self.lib.zko_gl_root_of_unity(k)

    def poseidon_permute(self, st, naive=False):
        a = np.array(st, dtype=np.uint64)
        (self.lib.zko_poseidon_permute_naive if naive else self.lib.zko_poseidon_permute)(_ptr(a))
        return a

    def poseidon_permute_batch(self, states):
        a = np.ascontiguousarray(states, dtype=np.uint64).copy()
        self.lib.zko_poseidon_permute_batch(_ptr(a), C.c_size_t(a.size // 12))
        return a

    def poseidon_witness_row(self, inp, timestamp=0, filt=1):
        a = np.array(inp, dtype=np.uint64)
        row = np.zeros(POSEIDON_COLS, dtype=np.uint64)
        self.lib.zko_poseidon_witness_row(_ptr(a), C.c_uint64(timestamp), C.c_int(filt), _ptr(row))
        return row

    def hash_no_pad(self, data):
        a = np.ascontiguousarray(data, dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        self.lib.zko_poseidon_hash_no_pad(_ptr(a), C.c_size_t(a.size), _ptr(out))
        return out

    def hash_or_noop(self, data):
        a = np.ascontiguousarray(data, dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        self.lib.zko_poseidon_hash_or_noop(_ptr(a), C.c_size_t(a.size), _ptr(out))
        return out

    def two_to_one(self, l, r):
        l = np.ascontiguousarray(l, dtype=np.uint64)
        r = np.ascontiguousarray(r, dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        self.lib.zko_poseidon_two_to_one(_ptr(l), _ptr(r), _ptr(out))
        return out

    def keccakf(self, st):
        a = np.array(st, dtype=np.uint64)
        self.lib.zko_keccakf(_ptr(a))
        return a

    def keccakf_batch(self, states):
        a = np.ascontiguousarray(states, dtype=np.uint64).copy()
        self.lib.zko_keccakf_batch(_ptr(a), C.c_size_t(a.size // 25))
        return a

    def keccak256(self, msg: bytes):
        out = (C.c_uint8 * 32)()
        self.lib.zko_keccak256(C.c_char_p(msg), C.c_size_t(len(msg)), out)
        return bytes(out)

    def keccak_sponge_trace(self, inputs, input_off, meta, log_n):
        inputs = np.ascontiguousarray(inputs, dtype=np.uint8)
        input_off = np.ascontiguousarray(input_off, dtype=np.uint64)
        meta = np.ascontiguousarray(meta, dtype=np.uint64)
        out = np.zeros(470 << log_n, dtype=np.uint64)
        self.lib.zko_keccak_sponge_trace.restype = C.c_size_t
        self.lib.zko_keccak_sponge_trace.argtypes = [C.c_void_p, u64p, u64p, C.c_size_t, C.c_uint, u64p]
        used = self.lib.zko_keccak_sponge_trace(inputs.ctypes.data_as(C.c_void_p), _ptr(input_off), _ptr(meta), input_off.size - 1,
                                                log_n, _ptr(out))
        if used == 0 and input_off.size > 1:
            raise RuntimeError("oracle keccak_sponge_trace: bad operations")
        return out, used

    def poseidon_sponge_trace(self, inputs, input_off, meta, log_n):
        inputs = np.ascontiguousarray(inputs, dtype=np.uint8)
        input_off = np.ascontiguousarray(input_off, dtype=np.uint64)
        meta = np.ascontiguousarray(meta, dtype=np.uint64)
        out = np.zeros(110 << log_n, dtype=np.uint64)
        self.lib.zko_poseidon_sponge_trace.restype = C.c_size_t
        self.lib.zko_poseidon_sponge_trace.argtypes = [C.c_void_p, u64p, u64p, C.c_size_t, C.c_uint, u64p]
        used = self.lib.zko_poseidon_sponge_trace(inputs.ctypes.data_as(C.c_void_p), _ptr(input_off), _ptr(meta), input_off.size - 1,
                                                  log_n, _ptr(out))
        if used == 0 and input_off.size > 1:
            raise RuntimeError("oracle poseidon_sponge_trace: bad operations")
        return out, used

    def poseidon_trace_inputs(self, inputs, timestamps, log_n):
        inputs = np.ascontiguousarray(inputs, dtype=np.uint64).reshape(-1, 12)
        ts = np.ascontiguousarray(timestamps, dtype=np.uint64)
        out = np.zeros(POSEIDON_COLS << log_n, dtype=np.uint64)
        self.lib.zko_poseidon_trace_inputs.argtypes = [u64p, u64p, C.c_size_t, C.c_uint, u64p]
        self.lib.zko_poseidon_trace_inputs(_ptr(inputs), _ptr(ts), len(inputs), log_n, _ptr(out))
        return out

    def sha_extend_trace(self, inputs, timestamps, log_n):
        inputs = np.ascontiguousarray(inputs, dtype=np.uint8).reshape(-1, 16)
        ts = np.ascontiguousarray(timestamps, dtype=np.uint64)
        out = np.zeros(78 << log_n, dtype=np.uint64)
        self.lib.zko_sha_extend_trace.argtypes = [C.c_void_p, u64p, C.c_size_t, C.c_uint, u64p]
        self.lib.zko_sha_extend_trace(inputs.ctypes.data, _ptr(ts), len(inputs), log_n, _ptr(out))
        return out

    def sha_extend_sponge_trace(self, w16, meta, log_n):
        w16 = np.ascontiguousarray(w16, dtype=np.uint32).reshape(-1, 16)
        meta = np.ascontiguousarray(meta, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(76 << log_n, dtype=np.uint64)
        self.lib.zko_sha_extend_sponge_trace.restype = C.c_size_t
        self.lib.zko_sha_extend_sponge_trace.argtypes = [C.c_void_p, u64p, C.c_size_t, C.c_uint, u64p]
        used = self.lib.zko_sha_extend_sponge_trace(w16.ctypes.data, _ptr(meta), len(w16), log_n, _ptr(out))
        if used == 0 and len(w16):
            raise RuntimeError("oracle sha_extend_sponge_trace: rounds do not fit")
        return out, used

    def _sha_compress_args(self, hx, w, meta):
        hx = np.ascontiguousarray(hx, dtype=np.uint32).reshape(-1, 8)
        w = np.ascontiguousarray(w, dtype=np.uint32).reshape(-1, 64)
        meta = np.ascontiguousarray(meta, dtype=np.uint64).reshape(-1, 8)
        assert len(hx) == len(w) == len(meta)
        return hx, w, meta

    def sha_compress_trace(self, hx, w, meta, log_n):
        hx, w, meta = self._sha_compress_args(hx, w, meta)
        out = np.zeros(224 << log_n, dtype=np.uint64)
        self.lib.zko_sha_compress_trace.restype = C.c_size_t
        self.lib.zko_sha_compress_trace.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_size_t, C.c_uint, u64p]
        used = self.lib.zko_sha_compress_trace(hx.ctypes.data, w.ctypes.data, _ptr(meta), len(hx), log_n, _ptr(out))
        if used == 0 and len(hx):
            raise RuntimeError("oracle sha_compress_trace: rounds do not fit")
        return out

    def sha_compress_sponge_trace(self, hx, w, meta, log_n):
        hx, w, meta = self._sha_compress_args(hx, w, meta)
        out = np.zeros(127 << log_n, dtype=np.uint64)
        self.lib.zko_sha_compress_sponge_trace.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_size_t, C.c_uint, u64p]
        self.lib.zko_sha_compress_sponge_trace(hx.ctypes.data, w.ctypes.data, _ptr(meta), len(hx), log_n, _ptr(out))
        return out

    # ---- NTT / commitment
    def ntt(self, cols, log_n, inverse=False, coset_shift=0):
        a = np.ascontiguousarray(cols, dtype=np.uint64).copy()
        self.lib.zko_ntt(_ptr(a), C.c_size_t(a.size >> log_n), log_n, int(inverse), C.c_uint64(coset_shift))
        return a

    def batch_from_values(self, values, ncols, log_n, rate_bits=2, cap_height=4):
        a = np.ascontiguousarray(values, dtype=np.uint64)
        return OracleBatch(self, self.lib.zko_batch_from_values(_ptr(a), ncols, log_n, rate_bits, cap_height),
                           ncols, log_n, rate_bits, cap_height)

    def batch_from_coeffs(self, coeffs, ncols, log_n, rate_bits=2, cap_height=4):
        a = np.ascontiguousarray(coeffs, dtype=np.uint64)
        return OracleBatch(self, self.lib.zko_batch_from_coeffs(_ptr(a), ncols, log_n, rate_bits, cap_height),
                           ncols, log_n, rate_bits, cap_height)

    # ---- STARK
    def standard_config(self):
        cfg = StarkConfig()
        self.lib.zko_standard_config(C.byref(cfg))
        return cfg

    def poseidon_trace(self, seed, num_perms, log_n):
        out = np.zeros(POSEIDON_COLS << log_n, dtype=np.uint64)
        self.lib.zko_poseidon_trace(C.c_uint64(seed), C.c_size_t(num_perms), log_n, _ptr(out))
        return out

    def debug_constraints(self, table_id, trace, ncols, log_n):
        """(constraints per row, first failing (row, constraint index) or None) on the plain trace rows."""
        br, bi = C.c_long(), C.c_long()
        self.lib.zko_debug_constraints.restype = C.c_long
        self.lib.zko_debug_constraints.argtypes = [C.c_int, u64p, C.c_size_t, C.c_uint, C.POINTER(C.c_long), C.POINTER(C.c_long)]
        n = self.lib.zko_debug_constraints(table_id, _ptr(np.ascontiguousarray(trace, dtype=np.uint64)), ncols, log_n, C.byref(br), C.byref(bi))
        return n, (None if br.value < 0 else (br.value, bi.value))

    def row_constraints(self, table_id, lv, nv, is_first=False, is_last=False):
        """Values of every constraint of table `table_id` on one (local, next) row pair."""
        self.lib.zko_debug_row_constraints.restype = C.c_long
        self.lib.zko_debug_row_constraints.argtypes = [C.c_int, u64p, u64p, C.c_int, C.c_int, u64p, C.c_size_t]
        out = np.zeros(8192, dtype=np.uint64)
        n = self.lib.zko_debug_row_constraints(table_id, _ptr(np.ascontiguousarray(lv, dtype=np.uint64)),
                                               _ptr(np.ascontiguousarray(nv, dtype=np.uint64)), int(is_first), int(is_last), _ptr(out), out.size)
        return out[:n]

    def constraint_evals(self, table_id, rows, log_witness, rate_bits, alpha):
        """stark_testing.rs:21-70: constraint values of `table_id` on the low-degree extension `rows` (ncols x size, natural order)."""
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        size = (1 << log_witness) << rate_bits
        assert rows.ndim == 2 and rows.shape[1] == size
        out = np.zeros(size, dtype=np.uint64)
        self.lib.zko_constraint_evals.restype = None
        self.lib.zko_constraint_evals.argtypes = [C.c_int, u64p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint64, u64p]
        self.lib.zko_constraint_evals(table_id, _ptr(rows), rows.shape[0], log_witness, rate_bits, int(alpha), _ptr(out))
        return out

    def poseidon_eval_row(self, row, alphas):
        row = np.ascontiguousarray(row, dtype=np.uint64)
        al = np.ascontiguousarray(alphas, dtype=np.uint64)
        out = np.zeros(al.size, dtype=np.uint64)
        self.lib.zko_poseidon_eval_row(_ptr(row), _ptr(al), C.c_size_t(al.size), _ptr(out))
        return out

    def quotient_poseidon(self, trace_b, aux_b, num_helpers, alphas):
        nh = (C.c_uint32 * len(num_helpers))(*num_helpers)
        al = np.ascontiguousarray(alphas, dtype=np.uint64)
        out = np.zeros(al.size * (2 << trace_b.log_n), dtype=np.uint64)
        self.lib.zko_quotient_poseidon(trace_b.h, aux_b.h, nh, len(num_helpers), _ptr(al), C.c_size_t(al.size), _ptr(out))
        return out

    def quotient(self, trace_b, aux_b, num_helpers, alphas, table_id=0):
        nh = (C.c_uint32 * len(num_helpers))(*num_helpers)
        al = np.ascontiguousarray(alphas, dtype=np.uint64)
        out = np.zeros(al.size * (2 << trace_b.log_n), dtype=np.uint64)
        self.lib.zko_quotient.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, u64p, C.c_size_t, u64p]
        self.lib.zko_quotient(table_id, trace_b.h, aux_b.h, nh, len(num_helpers), _ptr(al), al.size, _ptr(out))
        return out

    def proof_words(self, cfg, log_n, ncols, naux, nctl):
        return self.lib.zko_proof_words(C.byref(cfg), log_n, ncols, naux, nctl)

    def keccak_trace(self, inputs, timestamps, log_n):
        inputs = np.ascontiguousarray(inputs, dtype=np.uint64).reshape(-1, 25)
        ts = np.ascontiguousarray(timestamps, dtype=np.uint64)
        out = np.zeros(2431 << log_n, dtype=np.uint64)
        self.lib.zko_keccak_trace.restype = C.c_size_t
        self.lib.zko_keccak_trace.argtypes = [u64p, u64p, C.c_size_t, C.c_uint, u64p]
        used = self.lib.zko_keccak_trace(_ptr(inputs), _ptr(ts), len(inputs), log_n, _ptr(out))
        if used == 0 and len(inputs):
            raise RuntimeError("oracle keccak_trace: permutations do not fit")
        return out

    def logic_trace(self, ops, log_n):
        ops = np.ascontiguousarray(ops, dtype=np.uint32).reshape(-1, 3)
        out = np.zeros(69 << log_n, dtype=np.uint64)
        self.lib.zko_logic_trace.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, u64p]
        self.lib.zko_logic_trace(ops.ctypes.data, len(ops), log_n, _ptr(out))
        return out

    def prove(self, trace, log_n, aux, num_helpers, challenger=None, cfg=None, ncols=POSEIDON_COLS, want_stages=False, table_id=0):
        cfg = cfg or self.standard_config()
        ch = challenger or Challenger()
        naux = aux.size >> log_n
        nh = (C.c_uint32 * len(num_helpers))(*num_helpers)
        proof = np.zeros(self.proof_words(cfg, log_n, ncols, naux, len(num_helpers)), dtype=np.uint64)
        stages = (C.c_double * 8)()
        rc = self.lib.zko_prove_single_table(table_id, C.byref(cfg), _ptr(np.ascontiguousarray(trace)), ncols, log_n,
                                             _ptr(np.ascontiguousarray(aux)), naux, nh, len(num_helpers), C.byref(ch),
                                             _ptr(proof), stages)
        if rc != 0:
            raise RuntimeError("oracle prove_single_table failed: %d" % rc)
        return (proof, list(stages)) if want_stages else proof

    def verify(self, proof, naux, num_helpers, challenger=None, cfg=None, ncols=POSEIDON_COLS, table_id=0):
        cfg = cfg or self.standard_config()
        ch = challenger or Challenger()
        nh = (C.c_uint32 * len(num_helpers))(*num_helpers)
        return self.lib.zko_verify_single_table(table_id, C.byref(cfg), _ptr(np.ascontiguousarray(proof)), ncols, naux, nh,
                                                len(num_helpers), C.byref(ch))

    # ---- cross-table lookups / multi-table proofs (descriptor builders live in zkm_amd/ctl.py: pure marshalling)
    def _ctl_sigs(self):
        if getattr(self, "_ctl_ready", False):
            return
        L = self.lib
        vp = C.c_void_p
        L.zko_ctl_data.argtypes = [vp, vp, vp, C.c_size_t, u64p, C.c_size_t, C.c_uint, u64p]
        L.zko_lookup_helper_columns.argtypes = [vp, vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64, u64p, C.c_size_t, C.c_uint, u64p]
        L.zko_check_ctls.restype = C.c_int
        L.zko_check_ctls.argtypes = [vp, C.c_size_t, vp, vp, C.c_size_t]
        L.zko_prove_single_table_ctl.restype = C.c_int
        L.zko_prove_single_table_ctl.argtypes = [C.c_int, C.POINTER(StarkConfig), u64p, C.c_size_t, C.c_uint, u64p, C.c_size_t, vp, vp,
                                                 vp, C.c_size_t, u64p, C.POINTER(Challenger), u64p]
        L.zko_num_lookup_columns.restype = C.c_size_t
        L.zko_num_lookup_columns.argtypes = [C.c_int, C.POINTER(StarkConfig)]
        L.zko_verify_single_table_ctl.restype = C.c_int
        L.zko_verify_single_table_ctl.argtypes = [C.c_int, C.POINTER(StarkConfig), u64p, C.c_size_t, C.c_size_t, vp, vp, vp,
                                                  C.c_size_t, u64p, C.POINTER(Challenger)]
        L.zko_all_proof_words.restype = C.c_size_t
        L.zko_all_proof_words.argtypes = [C.POINTER(StarkConfig), vp, C.c_size_t, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.zko_prove_with_traces.restype = C.c_int
        L.zko_prove_with_traces.argtypes = [C.POINTER(StarkConfig), vp, C.c_size_t, vp, vp, C.c_size_t, u64p, C.c_size_t, u64p, u64p]
        L.zko_verify_all.restype = C.c_int
        L.zko_verify_all.argtypes = [C.POINTER(StarkConfig), vp, C.c_size_t, vp, vp, C.c_size_t, u64p, C.c_size_t, u64p, u64p]
        self._ctl_ready = True

    def ctl_data(self, ctl_table, zs, colset_ids, trace, ncols, log_n):
        self._ctl_sigs()
        naux = int(zs["num_helpers"].sum()) + len(zs)
        aux = np.zeros(naux << log_n, dtype=np.uint64)
        st = ctl_table.pack()
        self.lib.zko_ctl_data(C.addressof(st), zs.ctypes.data, colset_ids.ctypes.data, len(zs), _ptr(np.ascontiguousarray(trace)), ncols,
                              log_n, _ptr(aux))
        return aux

    def lookup_helper_columns(self, ctl_table, colset_ids, table_col, freq_col, challenge, trace, ncols, log_n):
        self._ctl_sigs()
        ids = np.ascontiguousarray(colset_ids, dtype=np.uint32)
        out = np.zeros(((len(ids) + 1) // 2 + 1) << log_n, dtype=np.uint64)
        st = ctl_table.pack()
        self.lib.zko_lookup_helper_columns(C.addressof(st), ids.ctypes.data, len(ids), table_col, freq_col, C.c_uint64(challenge),
                                           _ptr(np.ascontiguousarray(trace)), ncols, log_n, _ptr(out))
        return out

    def num_lookup_columns(self, table_id, cfg=None):
        self._ctl_sigs()
        cfg = cfg or self.standard_config()
        return self.lib.zko_num_lookup_columns(table_id, C.byref(cfg))

    def memory_trace(self, ops, log_n):
        """ops: nops x 6 (context, segment, virt, timestamp, is_read, value).  Returns (13 x 2^log_n trace, natural rows)."""
        ops = np.ascontiguousarray(ops, dtype=np.uint64).reshape(-1, 6)
        out = np.zeros(13 << log_n, dtype=np.uint64)
        self.lib.zko_memory_trace.restype = C.c_size_t
        self.lib.zko_memory_trace.argtypes = [u64p, C.c_size_t, C.c_uint, u64p]
        rows = self.lib.zko_memory_trace(_ptr(ops), len(ops), log_n, _ptr(out))
        if rows == 0:
            raise RuntimeError("oracle memory_trace: operations do not fit in 2^log_n rows (or a range check failed)")
        return out, rows

    def prove_ctl(self, trace, log_n, aux, ctl_table, zs, colset_ids, challenger=None, cfg=None, ncols=POSEIDON_COLS, table_id=0,
                  lookup_challenges=None):
        self._ctl_sigs()
        cfg = cfg or self.standard_config()
        ch = challenger or Challenger()
        naux = aux.size >> log_n
        lk = None if lookup_challenges is None else np.ascontiguousarray(lookup_challenges, dtype=np.uint64)
        proof = np.zeros(self.proof_words(cfg, log_n, ncols, naux + self.num_lookup_columns(table_id, cfg), len(zs)), dtype=np.uint64)
        st = ctl_table.pack()
        rc = self.lib.zko_prove_single_table_ctl(table_id, C.byref(cfg), _ptr(np.ascontiguousarray(trace)), ncols, log_n,
                                                 _ptr(np.ascontiguousarray(aux)), naux, C.addressof(st), zs.ctypes.data,
                                                 colset_ids.ctypes.data, len(zs), None if lk is None else _ptr(lk), C.byref(ch), _ptr(proof))
        if rc:
            raise RuntimeError("oracle prove_single_table_ctl failed: %d" % rc)
        return proof

    def verify_ctl(self, proof, naux, ctl_table, zs, colset_ids, challenger=None, cfg=None, ncols=POSEIDON_COLS, table_id=0,
                   lookup_challenges=None):
        self._ctl_sigs()
        cfg = cfg or self.standard_config()
        ch = challenger or Challenger()
        lk = None if lookup_challenges is None else np.ascontiguousarray(lookup_challenges, dtype=np.uint64)
        st = ctl_table.pack()
        return self.lib.zko_verify_single_table_ctl(table_id, C.byref(cfg), _ptr(np.ascontiguousarray(proof)), ncols, naux, C.addressof(st),
                                                    zs.ctypes.data, colset_ids.ctypes.data, len(zs), None if lk is None else _ptr(lk),
                                                    C.byref(ch))

    def _pack_all(self, tables, ctls):
        from zkm_amd import ctl as zc
        packed = [(tid, tr.ctypes.data, ncols, log_n, ct) for (tid, tr, ncols, log_n, ct) in tables]
        tarr, keep = zc.pack_tables(packed)
        carr, sides = zc.pack_ctls(ctls)
        return tarr, carr, sides, keep

    def check_ctls(self, tables, ctls):
        self._ctl_sigs()
        tarr, carr, sides, keep = self._pack_all(tables, ctls)
        return self.lib.zko_check_ctls(tarr, len(tables), carr.ctypes.data, sides.ctypes.data, len(carr))

    def all_proof_words(self, tables, ctls, cfg=None):
        self._ctl_sigs()
        cfg = cfg or self.standard_config()
        tarr, carr, sides, keep = self._pack_all(tables, ctls)
        offs = (C.c_size_t * (len(tables) + 1))()
        total = self.lib.zko_all_proof_words(C.byref(cfg), tarr, len(tables), carr.ctypes.data, sides.ctypes.data, len(carr), offs)
        return total, list(offs)

    def prove_with_traces(self, tables, ctls, public_values=(), cfg=None):
        """tables: list of (table_id, trace ndarray, ncols, log_n, CtlTable); ctls: list of (looking sides, looked side)."""
        self._ctl_sigs()
        cfg = cfg or self.standard_config()
        tarr, carr, sides, keep = self._pack_all(tables, ctls)
        total, offs = self.all_proof_words(tables, ctls, cfg)
        proofs = np.zeros(total, dtype=np.uint64)
        chal = np.zeros(2 * cfg.num_challenges, dtype=np.uint64)
        pub = np.ascontiguousarray(public_values, dtype=np.uint64)
        rc = self.lib.zko_prove_with_traces(C.byref(cfg), tarr, len(tables), carr.ctypes.data, sides.ctypes.data, len(carr), _ptr(pub),
                                            pub.size, _ptr(proofs), _ptr(chal))
        if rc:
            raise RuntimeError("oracle prove_with_traces failed: %d" % rc)
        return proofs, chal, offs

    def verify_all(self, tables, ctls, proofs, challenges, public_values=(), cfg=None):
        self._ctl_sigs()
        cfg = cfg or self.standard_config()
        tarr, carr, sides, keep = self._pack_all(tables, ctls)
        pub = np.ascontiguousarray(public_values, dtype=np.uint64)
        return self.lib.zko_verify_all(C.byref(cfg), tarr, len(tables), carr.ctypes.data, sides.ctypes.data, len(carr), _ptr(pub),
                                       pub.size, _ptr(np.ascontiguousarray(proofs)), _ptr(np.ascontiguousarray(challenges)))

    def prove_openings(self, tb, ab, qb, nctl_zs, challenger=None, cfg=None):
        cfg = cfg or self.standard_config()
        ch = challenger or Challenger()
        self.lib.zko_prove_openings.restype = C.c_int
        self.lib.zko_prove_openings.argtypes = [C.POINTER(StarkConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                                C.POINTER(Challenger), u64p]
        proof = np.zeros(self.proof_words(cfg, tb.log_n, tb.ncols, ab.ncols, nctl_zs), dtype=np.uint64)
        rc = self.lib.zko_prove_openings(C.byref(cfg), tb.h, ab.h, qb.h, nctl_zs, C.byref(ch), _ptr(proof))
        if rc:
            raise RuntimeError("oracle prove_openings failed: %d" % rc)
        return proof

    def verify_openings(self, proof, ncols, naux, nctl_zs, challenger=None, cfg=None):
        cfg = cfg or self.standard_config()
        ch = challenger or Challenger()
        self.lib.zko_verify_openings.restype = C.c_int
        self.lib.zko_verify_openings.argtypes = [C.POINTER(StarkConfig), u64p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(Challenger)]
        return self.lib.zko_verify_openings(C.byref(cfg), _ptr(np.ascontiguousarray(proof)), ncols, naux, nctl_zs, C.byref(ch))

    # ---- challenger
    def challenger(self):
        ch = Challenger()
        self.lib.zko_challenger_init(C.byref(ch))
        return ch

    def observe(self, ch, elems):
        a = np.ascontiguousarray(elems, dtype=np.uint64)
        self.lib.zko_challenger_observe(C.byref(ch), _ptr(a), C.c_size_t(a.size))

    def challenge(self, ch):
        return self.lib.zko_challenger_get(C.byref(ch))


class OracleBatch:
    def __init__(self, o, h, ncols, log_n, rate_bits, cap_height):
        self.o, self.h, self.ncols, self.log_n, self.rate_bits, self.cap_height = o, h, ncols, log_n, rate_bits, cap_height

    def __del__(self):
        if self.h:
            self.o.lib.zko_batch_free(self.h)
            self.h = None

    @property
    def lde_bits(self):
        return self.log_n + self.rate_bits

    def cap(self):
        out = np.zeros(4 << self.cap_height, dtype=np.uint64)
        self.o.lib.zko_batch_cap(self.h, _ptr(out))
        return out

    def coeffs(self):
        out = np.zeros(self.ncols << self.log_n, dtype=np.uint64)
        self.o.lib.zko_batch_coeffs(self.h, _ptr(out))
        return out

    def lde_row(self, natural_index):
        out = np.zeros(self.ncols, dtype=np.uint64)
        self.o.lib.zko_batch_lde_row(self.h, natural_index, _ptr(out))
        return out

    def leaf(self, i):
        out = np.zeros(self.ncols, dtype=np.uint64)
        self.o.lib.zko_batch_leaf(self.h, i, _ptr(out))
        return out

    def merkle_path(self, i):
        out = np.zeros(4 * (self.lde_bits - self.cap_height), dtype=np.uint64)
        self.o.lib.zko_batch_merkle_path(self.h, i, _ptr(out))
        return out

    def digest_layer(self, level):
        out = np.zeros(4 << (self.lde_bits - level), dtype=np.uint64)
        self.o.lib.zko_batch_digest_layer(self.h, level, _ptr(out))
        return out
