#!/bin/bash
# tools/gpu_r04_final.sh <tag> -- the round's ONE final pass on the final code: GPU suite, PMC passes that stamp pmc_latest.json, configs
# 4 / 5 with counters, an 8-rank rehearsal of the fallback path on the one GPU, the default bench line.
TAG=${1:-r04_z}
O=gpurun_out/$TAG
mkdir -p $O
export GPU_MAX_HW_QUEUES=16
timeout 1500 python -m pytest tests -m gpu -q -x --durations=6 > $O/pytest.log 2>&1; tail -10 $O/pytest.log
bash tools/collect_pmc.sh $TAG > $O/collect.log 2>&1; tail -24 $O/collect.log
cp gpurun_out/prof_$TAG/pmc_latest.json profiles/pmc_latest.json      # (on the GPU box: the bench line below quotes the traffic of THIS code)
bash tools/gpu_configs.sh ${TAG}_cfg > $O/configs.log 2>&1; tail -30 $O/configs.log
ZKM_BENCH_SHARE_GPU=1 ZKM_RCCL_PROBE_TIMEOUT_S=120 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29621 \
    bench.py --gpus 8 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --contexts 1 > $O/rehearsal_8ranks_1gpu.json 2> $O/rehearsal_8ranks_1gpu.err
echo "8 ranks 1 gpu rc=$?"; grep -c "zkm preflight" $O/rehearsal_8ranks_1gpu.err; tail -c 400 $O/rehearsal_8ranks_1gpu.json
ZKM_FORCE_PG=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29623 \
    bench.py --gpus 1 --steps 4 --warmup 1 --no-extras --no-cpu-baseline > $O/rccl_world1.json 2> $O/rccl_world1.err; echo "rccl world1 rc=$?"
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
print("kernels", d["kernel_ms_per_proof"])
print("seg16", {k: d["segment_2_16"].get(k) for k in ("ms_per_segment", "launches_per_segment")}, [(c["contexts"], round(c["segments_per_s"], 1)) for c in d["segment_2_16"].get("concurrent", [])])
print("fri", json.dumps(d.get("fri_2_22", {}).get("per_kernel_hbm")))
print("cpu", d.get("cpu_baseline", {}).get("value"), "errors", {x: d[x] for x in d if x.endswith("_error")})
P
