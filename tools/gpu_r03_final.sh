#!/bin/bash
# last run of the round on the final code: the GPU suite, the default bench line, the PMC passes that stamp profiles/pmc_latest.json
TAG=${1:-r03_h}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest.log 2>&1; tail -9 $O/pytest.log
bash tools/collect_pmc.sh $TAG > $O/collect.log 2>&1; tail -24 $O/collect.log
cp gpurun_out/prof_$TAG/pmc_latest.json profiles/pmc_latest.json      # (on the GPU box: so that the bench line below quotes the traffic of THIS code)
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python tools/bench_segment.py 16 > $O/seg16.json 2> $O/seg16.err
rm -rf gpurun_out/prof_$TAG/trace
find gpurun_out/prof_$TAG -name "*counter_collection.csv" -size +20M -delete
python - <<P
import json
d = json.load(open("$O/bench.json"))
print("value", round(d["value"], 3), "ms/step", round(d["ms_per_step"], 2), "single", round(d["single_context"]["ms_per_step"], 2))
print("roofline", d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["valu_issue"].get("at_measured_clock"))
print("ntt", {k: d["roofline_ntt"][k] for k in ("frac", "traffic", "ms_per_proof")})
print("errors", {x: d[x] for x in d if x.endswith("_error")})
P
