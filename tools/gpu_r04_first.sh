#!/bin/bash
# tools/gpu_r04_first.sh <tag> -- round 4, first GPU pass: the GPU suite on the refactored host code, the RCCL rehearsals on the one
# reachable GPU (world 1 through torchrun with ZKM_FORCE_PG; two ranks sharing the GPU, where RCCL must refuse and gloo take over),
# the crowded-host experiment, and a default bench line as this round's baseline.
TAG=${1:-r04_a}
O=gpurun_out/$TAG
mkdir -p $O
export GPU_MAX_HW_QUEUES=16
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest.log 2>&1; tail -12 $O/pytest.log
# (a) RCCL at world 1, as the driver launches N > 1: torchrun, backend nccl, probe child + in-process group + barrier + all_reduce(MAX)
ZKM_FORCE_PG=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 1 --steps 4 --warmup 1 --no-extras --no-cpu-baseline > $O/rccl_world1.json 2> $O/rccl_world1.err
echo "rccl world1 rc=$?"; grep "zkm preflight\|RCCL" $O/rccl_world1.err | tail -3
# (b) two ranks on ONE GPU: RCCL refuses a communicator with a duplicate device -> the fallback runs against the real library
ZKM_BENCH_SHARE_GPU=1 ZKM_RCCL_PROBE_TIMEOUT_S=90 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 \
    bench.py --gpus 2 --steps 2 --warmup 1 --no-extras --no-cpu-baseline --contexts 2 > $O/rccl_2ranks_1gpu.json 2> $O/rccl_2ranks_1gpu.err
echo "2 ranks 1 gpu rc=$?"; grep "zkm preflight\|RCCL" $O/rccl_2ranks_1gpu.err | tail -4
python - <<P
import json
for f in ("rccl_world1", "rccl_2ranks_1gpu"):
    try:
        d = json.loads([l for l in open("$O/%s.json" % f) if l.startswith("{")][-1])
        print(f, d["value"], d["config"]["process_group"], json.dumps(d["config"]["process_group_detail"]))
    except Exception as e:
        print(f, "FAILED", e)
P
timeout 1500 python tools/crowded_host.py > $O/crowded_host.txt 2> $O/crowded_host.err; tail -16 $O/crowded_host.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
python - <<P
import json
d = json.load(open("$O/bench.json"))
print("value", round(d["value"], 3), "ms/step", round(d["ms_per_step"], 2), "single", round(d["single_context"]["ms_per_step"], 2))
print("kernels", d["kernel_ms_per_proof"])
print("seg16", {k: d["segment_2_16"].get(k) for k in ("ms_per_segment", "launches_per_segment", "segments_per_s")}, d["segment_2_16"].get("concurrent"))
print("cpu", d.get("cpu_baseline"))
print("fri", json.dumps(d.get("fri_2_22", {}).get("per_kernel_hbm")))
print("sponge", {k: d.get("keccak_sponge_2_20", {}).get(k) for k in ("witness_ms", "witness_kernel_ms", "witness_ms_host_inputs", "commit_ms")})
print("errors", {x: d[x] for x in d if x.endswith("_error")})
P
