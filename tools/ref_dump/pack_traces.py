#!/usr/bin/env python3
"""reference_all_proof_traces.bin (ref_dump.rs::ref_dump_all_proof, ~10 MB of mostly small field elements) -> a compressed .npz next
to it, small enough to commit under tests/golden/ as a fixture (data only: inputs of the reference's own prover run).

  python tools/ref_dump/pack_traces.py tests/golden/reference_all_proof_traces.bin      -> tests/golden/reference_all_proof_traces.npz
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_self_fixtures import read_traces  # noqa: E402

if __name__ == "__main__":
    src = sys.argv[1]
    traces, ncols, log_n = read_traces(src)
    dst = os.path.splitext(src)[0] + ".npz"
    np.savez_compressed(dst, ncols=np.array(ncols), log_n=np.array(log_n), **{"t%d" % i: t for i, t in enumerate(traces)})
    print("%s: %d tables, %.1f MB -> %.1f MB" % (dst, len(traces), os.path.getsize(src) / 1e6, os.path.getsize(dst) / 1e6))
