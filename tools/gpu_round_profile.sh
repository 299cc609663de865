#!/bin/bash
# tools/gpu_round_profile.sh <tag> -- one gpurun call that produces what profiles/<tag>_* holds: bench line with 4 contexts per GPU, rocprofv3 kernel stats + PMC passes (single context), segment benchmark, GPU suite
mkdir -p gpurun_out/round
python bench.py --steps 8 --warmup 1 > gpurun_out/round/bench.json 2> gpurun_out/round/bench.err
bash tools/collect_pmc.sh ${1:-r02_d} > gpurun_out/round/collect.log 2>&1
tail -30 gpurun_out/round/collect.log
python tools/bench_segment.py 20 > gpurun_out/round/seg20.json 2>&1
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/round/pytest.log 2>&1; tail -3 gpurun_out/round/pytest.log
