cd "${GRAFT_REPO_ROOT:-/root/repo}"; export GPU_MAX_HW_QUEUES=16
: > gpurun_out/r04_tp_pow.txt
for t in "throughput_profile=1" "throughput_profile=1,pow_round_log=15" "throughput_profile=1,pow_round_log=14" "throughput_profile=1,pow_round_log=13" "throughput_profile=1" "throughput_profile=1,pow_round_log=14"; do
  ZKM_SEG_TUNING="$t" timeout 300 python tools/auxpipe_ab.py 6 16 >> gpurun_out/r04_tp_pow.txt 2>&1
done
cat gpurun_out/r04_tp_pow.txt
