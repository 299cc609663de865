#!/bin/bash
# tools/gpu_final.sh <tag> -- a round's final pass on the final code, ONE gpurun call: GPU suite, the PMC passes that stamp
# profiles/pmc_latest.json, configs 4 / 5 with counters, the lock-step segment profile (kernel stats + VALU budget), the default bench line.
# Everything lands under gpurun_out/<tag>/; copy what the judge should read into profiles/ afterwards (tools/collect_profiles.py <tag>).
TAG=${1:-r05_z}
O=gpurun_out/$TAG
mkdir -p $O
export GPU_MAX_HW_QUEUES=16
timeout 1700 python -m pytest tests -m gpu -q -x --durations=6 > $O/pytest.log 2>&1; tail -10 $O/pytest.log
bash tools/collect_pmc.sh $TAG > $O/collect.log 2>&1; tail -24 $O/collect.log
cp gpurun_out/prof_$TAG/pmc_latest.json profiles/pmc_latest.json      # (on the GPU box: the bench line below quotes the traffic of THIS code)
bash tools/gpu_configs.sh ${TAG}_cfg > $O/configs.log 2>&1; tail -30 $O/configs.log
bash tools/gpu_lockstep_prof.sh $TAG 16 > $O/lockstep_prof.log 2>&1; tail -12 $O/lockstep_prof.log
cp gpurun_out/lockstep_valu_latest.json profiles/lockstep_valu_latest.json
T0=$SECONDS; python bench.py > $O/bench.json 2> $O/bench.err; echo "bench.py wall $((SECONDS - T0)) s"; tail -c 300 $O/bench.err
python tools/bench_segment.py 16 > $O/seg16.json 2> $O/seg16.err
rm -rf gpurun_out/prof_$TAG/trace
find gpurun_out/prof_$TAG -name "*counter_collection.csv" -size +20M -delete
python - <<P
import json
d = json.load(open("$O/bench.json"))
print("value", round(d["value"], 3), "ms/step", round(d["ms_per_step"], 2), "single", round(d["single_context"]["ms_per_step"], 2))
print("roofline", {k: d["roofline"].get(k) for k in ("bound", "frac", "traffic")})
print("ntt", {k: d["roofline_ntt"][k] for k in ("frac", "traffic", "ms_per_proof")})
print("kernels", d["kernel_ms_per_proof"])
s = d.get("segment_2_16", {})
print("seg16", {k: s.get(k) for k in ("ms_per_segment", "launches_per_segment")}, "lockstep", [(c.get("contexts"), c.get("segments_per_call"), round(c.get("segments_per_s", 0), 1), c.get("valu_budget_frac")) for c in s.get("lockstep", []) if "contexts" in c])
print("fri", json.dumps(d.get("fri_2_22", {}).get("per_kernel_hbm")))
print("cpu", d.get("cpu_baseline", {}).get("value"), "errors", {x: d[x] for x in d if x.endswith("_error")})
P
