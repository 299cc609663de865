#!/bin/bash
# tools/gpu_r03_configs.sh <tag> -- BASELINE configs 3, 4, 5 on one MI355X (run through gpurun): the config-3 GPU test + the 64-segment
# bench line, then configs 4 / 5 timed alone and under rocprofv3 (kernel trace, then FETCH_SIZE / WRITE_SIZE in their own passes).
TAG=${1:-r03_a}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout 1500 python -m pytest tests/test_config3.py tests/test_gpu_large_parity.py::test_proof_is_bit_exact_2_20 -m gpu -q -x -s > $O/pytest_config3.log 2>&1; tail -5 $O/pytest_config3.log
cp gpurun_out/cpu_oracle_full_size.json $O/ 2>/dev/null
python bench.py --segments 64 --no-extras --no-cpu-baseline > $O/bench_segments64.json 2> $O/bench_segments64.err; tail -c 600 $O/bench_segments64.json
python tools/bench_configs.py all > $O/configs.json 2> $O/configs.err; tail -c 1500 $O/configs.json
cd /tmp && export TMPDIR=/tmp
for W in fri sponge; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$W -- python $R/tools/bench_configs.py $W > $O/trace_$W.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $O/pmc_${W}_$C -- python $R/tools/bench_configs.py $W > $O/pmc_${W}_$C.log 2>&1
  done
  find $O/trace_$W -name "*kernel_stats.csv" -exec cp {} $O/${W}_kernel_stats.csv \;
done
cd $R
python tools/summarize_config_pmc.py $O > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt | tail -40
# keep the merged output small: drop the raw traces, keep the digests
rm -rf $O/trace_fri $O/trace_sponge
find $O -name "*counter_collection.csv" -size +20M -delete
