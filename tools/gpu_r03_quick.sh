#!/bin/bash
# development round trip: GPU parity subset, single-context bench + rocprofv3 kernel stats, configs 4/5
TAG=${1:-r03_q}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_prove.py tests/test_segment.py tests/test_gpu_tables.py tests/test_gpu_ctl.py "tests/test_gpu_large_parity.py::test_large_commit_matches_oracle" -m gpu -q -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python tools/bench_configs.py sponge > $O/sponge.json 2> $O/sponge.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --steps 2 --warmup 1 --contexts 1 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
cd $R
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/trace
python - <<P
import json, csv
d = json.load(open("$O/bench.json"))
print("value", round(d["value"], 3), "ms/step", round(d["ms_per_step"], 2), "single", round(d["single_context"]["ms_per_step"], 2))
print("kernel", d["kernel_ms_per_proof"])
s = d.get("segment_2_16", {})
print("seg16", {k: s.get(k) for k in ("ms_per_segment", "kernel_ms_per_segment", "wall_over_kernel_sum", "launches_per_segment")}, s.get("stage_ms"), [(c["contexts"], round(c["segments_per_s"], 1)) for c in s.get("concurrent", [])])
print("host", d.get("host_resident"))
k = json.load(open("$O/sponge.json"))["keccak_sponge_2_20"]
print("sponge", {x: k.get(x) for x in ("witness_ms", "witness_kernel_ms", "witness_permutations_per_s", "commit_ms")}, k.get("keccakf_batch"))
print({x: d[x] for x in d if x.endswith("_error")})
for r in sorted(csv.DictReader(open("$O/kernel_stats.csv")), key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("%-70s calls %5s avg %9.1f us total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
P
