#!/bin/bash
# dev batch: microbenchmarks + timing + bench (outputs under gpurun_out/)
mkdir -p gpurun_out/b1
tools/ubench_issue > gpurun_out/b1/ubench_issue.json 2> gpurun_out/b1/ubench_issue.err
tools/ubench_strided > gpurun_out/b1/ubench_strided.txt 2>&1
ZKO_TIMING=1 python tools/time_full_segment.py > gpurun_out/b1/time_full_segment.txt 2>&1
python bench.py --steps 5 --warmup 1 > gpurun_out/b1/bench.json 2> gpurun_out/b1/bench.err
python -m pytest tests/test_gpu_large_parity.py tests/test_gpu_primitives.py -m gpu -q -x > gpurun_out/b1/pytest.log 2>&1
tail -3 gpurun_out/b1/pytest.log
head -c 600 gpurun_out/b1/bench.json
