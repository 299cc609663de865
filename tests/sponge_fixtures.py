"""Seeded synthetic Keccak-sponge operations (BASELINE config 5: lengths uniform in [1, 1088) bytes)."""
import numpy as np


def make_ops(seed, nops, max_len=1088):
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, max_len, nops)
    off = np.zeros(nops + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    data = rng.integers(0, 256, int(off[-1]), dtype=np.uint8)
    meta = np.zeros((nops, 4), dtype=np.uint64)
    meta[:, 0] = 0                      # context
    meta[:, 1] = 3                      # segment
    meta[:, 2] = rng.integers(0, 1 << 24, nops)  # virt base
    meta[:, 3] = np.arange(nops) * 7 + 1         # timestamp
    rows = int(np.sum(lens // 136 + 1))
    return data, off, meta.reshape(-1), rows


def ops_for_rows(seed, target_rows):
    """As many operations as fit in target_rows rows."""
    nops = max(1, int(target_rows / 4.6))
    while True:
        data, off, meta, rows = make_ops(seed, nops)
        if rows <= target_rows:
            return data, off, meta, rows, nops
        nops = int(nops * 0.98)
