"""CPU suite: the C-ABI library loads and exports every symbol include/zkm_hip.h declares; host-side
transcript code (no GPU needed) agrees with the oracle."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "zkm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zkm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(zkm):
    lib = zkm.load()
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libzkmhip.so does not export %s" % n
    assert set(names) == set(zkm.EXPORTS), set(names) ^ set(zkm.EXPORTS)
    assert b"gfx950" in lib.zkm_version()


def test_no_gpu_fails_loudly(zkm):
    import torch
    if torch.cuda.is_available():
        return
    try:
        zkm.Context(0)
    except zkm.ZkmError as e:
        assert "hip" in str(e).lower()
    else:
        raise AssertionError("Context creation must fail without a GPU (no CPU fallback)")


def test_host_challenger_matches_oracle(zkm, oracle):
    rng = np.random.default_rng(11)
    a, b = zkm.challenger_new(), oracle.challenger()
    for step in range(40):
        k = int(rng.integers(0, 20))
        elems = rng.integers(0, zkm.P, k, dtype=np.uint64)
        zkm.challenger_observe(a, elems)
        oracle.observe(b, elems)
        for _ in range(int(rng.integers(0, 11))):
            assert zkm.challenger_get(a) == oracle.challenge(b)
    assert list(a.state) == list(b.state)


def test_proof_layout_sizes_agree(zkm, oracle):
    cfg_o = oracle.standard_config()
    cfg = zkm.StarkConfig()
    zkm.load().zkm_standard_config(C.byref(cfg))
    for f, _ in cfg._fields_:
        assert getattr(cfg, f) == getattr(cfg_o, f)
    lib = zkm.load()
    for log_n in (5, 7, 12, 16, 20, 22):
        assert lib.zkm_proof_words(C.byref(cfg), log_n, 262, 4, 2) == oracle.proof_words(cfg_o, log_n, 262, 4, 2)
