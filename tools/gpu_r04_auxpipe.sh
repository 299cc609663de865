#!/bin/bash
# Pipelined auxiliary commitments: parity first, then A/B of one segment's latency and of the 8-context rate.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export GPU_MAX_HW_QUEUES=16
O=gpurun_out/r04_auxpipe.txt; : > $O
timeout 900 python -m pytest tests/test_segment.py -m gpu -x -q 2>&1 | tail -5 >> $O
for t in "aux_pipeline=0" "aux_pipeline=1" "aux_pipeline=1,commit_lanes=3" "aux_pipeline=1,commit_lanes=6" "aux_pipeline=1,commit_lanes=8" "aux_pipeline=0" "aux_pipeline=1"; do
  ZKM_SEG_TUNING="$t" timeout 300 python tools/auxpipe_ab.py 30 1 >> $O 2>&1
done
# (the "prio" lines of profiles/r04_segment_latency_round2.txt: context stream at the highest, lane streams at the lowest priority -- an experiment
#  behind an environment variable that was removed again after it showed no effect)
for t in "aux_pipeline=0" "aux_pipeline=1" "aux_pipeline=0,commit_lanes=2" "aux_pipeline=1,commit_lanes=2"; do
  ZKM_SEG_TUNING="$t" timeout 300 python tools/auxpipe_ab.py 6 8 >> $O 2>&1
done
ZKM_SEG_TUNING="throughput_profile=1" timeout 300 python tools/auxpipe_ab.py 5 16 >> $O 2>&1
cat $O
