#!/bin/bash
# round-2 profiles: rocprofv3 kernel stats + PMC passes of the bench command, the bench line itself, the segment benchmark
mkdir -p gpurun_out/b9
python bench.py --steps 5 --warmup 1 > gpurun_out/b9/bench.json 2> gpurun_out/b9/bench.err
bash tools/collect_pmc.sh r02_b > gpurun_out/b9/collect.log 2>&1
tail -30 gpurun_out/b9/collect.log
python tools/bench_segment.py 20 > gpurun_out/b9/seg20.json 2>&1
tail -5 gpurun_out/b9/seg20.json
