"""CPU suite: the N>1 sharding path under torch.distributed with the gloo backend, world_size 2."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import sys, time
    sys.path.insert(0, %r)
    import numpy as np
    from zkm_amd import dist as zd
    import torch.distributed as dist
    world, rank, _ = zd.init("gloo")
    assert world == 2
    NSEG = 7
    mine = zd.assign_segments(NSEG, world, rank)
    def fake_prove(s):            # stands in for Context.prove_single_table (no GPU in this suite)
        time.sleep(0.02 * (rank + 1))
        return np.full(4, 1000 + s, dtype=np.uint64)
    proofs, elapsed = zd.prove_segments(fake_prove, NSEG)
    slowest = 0.02 * 2 * len(zd.assign_segments(NSEG, world, 1))
    assert elapsed >= min(slowest, 0.02 * len(zd.assign_segments(NSEG, world, 0))) - 1e-3
    assert abs(zd.max_over_ranks(float(rank)) - 1.0) < 1e-9
    if rank == 0:
        assert sorted(proofs) == list(range(NSEG))
        assert all(int(proofs[s][0]) == 1000 + s for s in proofs)
        print("OK", mine, round(elapsed, 3))
    else:
        assert proofs is None
    dist.destroy_process_group()
""") % ROOT


def test_assign_segments_partition():
    sys.path.insert(0, ROOT)
    from zkm_amd.dist import assign_segments
    for world in (1, 2, 4, 8):
        seen = sorted(s for r in range(world) for s in assign_segments(64, world, r))
        assert seen == list(range(64))
        assert all(len(assign_segments(64, world, r)) == 64 // world for r in range(world))


def test_two_rank_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29513", str(script)], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK [0, 2, 4, 6]" in r.stdout
