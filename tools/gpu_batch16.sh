#!/bin/bash
# round-2 (c) profiles: bench line with 4 contexts per GPU, rocprofv3 kernel stats + PMC passes (single context), segment benchmark, GPU suite
mkdir -p gpurun_out/b16
python bench.py --steps 8 --warmup 1 > gpurun_out/b16/bench.json 2> gpurun_out/b16/bench.err
bash tools/collect_pmc.sh r02_c > gpurun_out/b16/collect.log 2>&1
tail -30 gpurun_out/b16/collect.log
python tools/bench_segment.py 20 > gpurun_out/b16/seg20.json 2>&1
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/b16/pytest.log 2>&1; tail -3 gpurun_out/b16/pytest.log
