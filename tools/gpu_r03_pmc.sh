#!/bin/bash
# development round trip: NTT / commitment parity subset, headline bench, rocprofv3 / PMC passes of the headline workload
TAG=${1:-r03_d}
O=gpurun_out/$TAG
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_large_parity.py::test_large_commit_matches_oracle tests/test_gpu_large_parity.py::test_prove_openings_bit_exact tests/test_gpu_prove.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
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
