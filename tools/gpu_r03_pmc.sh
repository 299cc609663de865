#!/bin/bash
# headline bench + the rocprofv3 / PMC passes of the headline workload only
TAG=${1:-r03_d}
O=gpurun_out/$TAG
mkdir -p $O
python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_noextras.json 2> $O/bench.err
bash tools/collect_pmc.sh $TAG > $O/collect.log 2>&1; tail -26 $O/collect.log
rm -rf gpurun_out/prof_$TAG/trace
find gpurun_out/prof_$TAG -name "*counter_collection.csv" -size +20M -delete
python - <<P
import json
d = json.load(open("$O/bench_noextras.json"))
print("value", round(d["value"], 3), "ms/step", round(d["ms_per_step"], 2), "single", round(d["single_context"]["ms_per_step"], 2))
print(d["kernel_ms_per_proof"])
P
