#!/bin/bash
# tools/gpu_round_profile.sh <tag> -- one gpurun call that produces what profiles/<tag>_* holds: the GPU suite, the bench line (4 contexts
# per GPU, all extras), rocprofv3 kernel stats + PMC passes of the headline (single context), BASELINE config 3 on one GPU
# (--segments 64), configs 4 / 5 with their own rocprofv3 + PMC passes, and the 2^20-cycle twelve-table segment.
TAG=${1:-r03_c}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest.log 2>&1; tail -14 $O/pytest.log
cp gpurun_out/cpu_oracle_full_size.json $O/ 2>/dev/null
cp gpurun_out/cpu_oracle_full_size.json profiles/r03_cpu_oracle_full_size.json 2>/dev/null   # (bench.py quotes it in stage_ms / cpu_baseline)
python bench.py --steps 8 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
bash tools/collect_pmc.sh $TAG > $O/collect.log 2>&1; tail -24 $O/collect.log
python bench.py --segments 64 --no-extras --no-cpu-baseline > $O/bench_segments64.json 2> $O/bench_segments64.err
python tools/bench_configs.py all > $O/configs.json 2> $O/configs.err
cd /tmp && export TMPDIR=/tmp
for W in fri sponge; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$W -- python $R/tools/bench_configs.py $W > $O/trace_$W.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $O/pmc_${W}_$C -- python $R/tools/bench_configs.py $W > $O/pmc_${W}_$C.log 2>&1
  done
  find $O/trace_$W -name "*kernel_stats.csv" -exec cp {} $O/${W}_kernel_stats.csv \;
done
cd $R
python tools/summarize_config_pmc.py $O > $O/pmc_configs_summary.txt 2>&1; tail -30 $O/pmc_configs_summary.txt
python tools/bench_segment.py 20 > $O/seg20.json 2> $O/seg20.err
python tools/bench_segment.py 16 > $O/seg16.json 2> $O/seg16.err
rm -rf $O/trace_fri $O/trace_sponge gpurun_out/prof_$TAG/trace
find $O gpurun_out/prof_$TAG -name "*counter_collection.csv" -size +20M -delete
python - <<P
import json
d = json.load(open("$O/bench.json"))
print("value", round(d["value"], 3), "ms/step", round(d["ms_per_step"], 2), "single", round(d["single_context"]["ms_per_step"], 2))
print("roofline", d["roofline"]["frac"], d["roofline"]["traffic"], "ntt", d["roofline_ntt"])
print("errors", {x: d[x] for x in d if x.endswith("_error")})
b = json.load(open("$O/bench_segments64.json"))
print("segments64", b["value"], b["ms_per_step"])
P
