#!/usr/bin/env python3
"""tests/golden/segment12.npz: the traces of the twelve-table test segment (tests/cpu_fixtures.build_full_segment: the sample MIPS
program plus Keccak / Poseidon / SHA precompile rows), as data.  Lets bench.py and tools/bench_segment.py time a whole
AllStark segment without importing the fixture builders or the oracle.  Keys: t0..t11 = table t of Table::all() order
(column-major uint64), log_n = the twelve heights.  ~0.3 MB compressed (mostly padding rows)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.oracle_py import Oracle  # noqa: E402
from tests import cpu_fixtures as CF  # noqa: E402

tables, ctls = CF.build_full_segment(Oracle())
np.savez_compressed(os.path.join(HERE, "segment12.npz"), log_n=np.array([t[3] for t in tables], dtype=np.int64),
                    **{"t%d" % i: np.asarray(t[1], dtype=np.uint64) for i, t in enumerate(tables)})
print("wrote segment12.npz", [(t[0], t[2], t[3]) for t in tables])
