#!/usr/bin/env python3
"""What does a latency-bound launch lose when a throughput-bound one shares the GPU?  (2^16-cycle segments: the Keccak table's leaf
sponge -- 2431 columns x 2^13 LDE rows = 304 dependent permutations per row in the four-lane form -- is the long pole of the trace
commitments while the CPU table's leaf hashing, one lane per row, fills the machine next to it: profiles/r04_segment_timeline.txt.)

Context K commits a Keccak-shaped table (2431 x 2^11) over and over, context C a CPU-shaped one (259 x 2^16); K's leaf kernel time comes
from the library's HIP-event profile.  Cases: K alone; K next to C; the same with the two contexts confined to disjoint halves of the
CUs (ZKM_CU_MASK_PART, csrc/core.hip).      python tools/contention_test.py  -> one JSON line per case
"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np  # noqa: E402

import zkm_amd  # noqa: E402

P = 0xFFFFFFFF00000001


def case(name, mask_k, mask_c, with_c, reps=4):
    def make(mask):
        if mask is None:
            os.environ.pop("ZKM_CU_MASK_PART", None)
        else:
            os.environ["ZKM_CU_MASK_PART"] = mask
        c = zkm_amd.Context(0)
        os.environ.pop("ZKM_CU_MASK_PART", None)
        return c
    ck, cc = make(mask_k), make(mask_c)
    rng = np.random.default_rng(1)
    vk = ck.alloc(2431 << 11).upload(rng.integers(0, P, 2431 << 11, dtype=np.uint64))
    vc = cc.alloc(259 << 16).upload(rng.integers(0, P, 259 << 16, dtype=np.uint64))
    zkm_amd.PolynomialBatch.from_values(ck, vk, 2431, 11).free()
    zkm_amd.PolynomialBatch.from_values(cc, vc, 259, 16).free()
    stop = threading.Event()
    count = [0]

    def loop_c():
        while not stop.is_set():
            zkm_amd.PolynomialBatch.from_values(cc, vc, 259, 16).free()
            count[0] += 1
    th = threading.Thread(target=loop_c)
    if with_c:
        th.start()
        time.sleep(0.05)
    ck.profile(True)
    ck.profile_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        zkm_amd.PolynomialBatch.from_values(ck, vk, 2431, 11).free()
    ck.synchronize()
    dt = (time.perf_counter() - t0) / reps
    rec = ck.profile_records()
    ck.profile(False)
    stop.set()
    if with_c:
        th.join()
    out = {"case": name, "keccak_commit_ms": dt * 1e3, "keccak_leaves_ms": rec["merkle_leaves"][1] / rec["merkle_leaves"][0],
           "us_per_absorb_step": rec["merkle_leaves"][1] / rec["merkle_leaves"][0] / 304 * 1e3,
           "keccak_kernels_ms": {k: round(v[1] / reps, 3) for k, v in rec.items()}, "cpu_table_commits_meanwhile": count[0]}
    print(json.dumps(out), flush=True)
    vk.free(); vc.free(); ck.close(); cc.close()


if __name__ == "__main__":
    case("keccak table alone", None, None, False)
    case("next to the CPU table's commitment (shared CUs)", None, None, True)
    case("disjoint halves of the CUs", "0/2", "1/2", True)
    case("keccak alone on half of the CUs", "0/2", None, False)
