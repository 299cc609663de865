#!/bin/bash
# tools/gpu_configs.sh <tag> -- BASELINE configs 4 and 5 on one MI355X at the code that is checked out: timed alone, then under
# rocprofv3 (kernel trace, then FETCH_SIZE / WRITE_SIZE in their own --pmc passes), digested by tools/summarize_config_pmc.py into
# <out>/{fri,sponge}_pmc.json (stamped with bench.py's code fingerprint; tools/bench_configs.py quotes them per kernel).
TAG=${1:-r04_e}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
export GPU_MAX_HW_QUEUES=16
python tools/bench_configs.py all > $O/configs.json 2> $O/configs.err; tail -c 600 $O/configs.json
cd /tmp && export TMPDIR=/tmp
for W in fri sponge; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$W -- python $R/tools/bench_configs.py $W > $O/trace_$W.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $O/pmc_${W}_$C -- python $R/tools/bench_configs.py $W > $O/pmc_${W}_$C.log 2>&1
  done
  find $O/trace_$W -name "*kernel_stats.csv" -exec cp {} $O/${W}_kernel_stats.csv \;
done
cd $R
python tools/summarize_config_pmc.py $O > $O/pmc_summary.txt 2>&1; tail -40 $O/pmc_summary.txt
# (on the GPU box: so that the second bench_configs run quotes the counters of THIS code)
cp $O/fri_pmc.json profiles/fri_2_22_pmc.json; cp $O/sponge_pmc.json profiles/keccak_sponge_2_20_pmc.json
python tools/bench_configs.py all > $O/configs_with_counters.json 2>> $O/configs.err
python -c "
import json; d=json.load(open('$O/configs_with_counters.json')); print(json.dumps(d['fri_2_22']['per_kernel_hbm'], indent=1)); print({k: d['keccak_sponge_2_20'][k] for k in ('witness_ms','witness_kernel_ms','witness_ms_host_inputs','commit_ms')})"
rm -rf $O/trace_fri $O/trace_sponge
find $O -name "*counter_collection.csv" -size +20M -delete
