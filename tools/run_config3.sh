#!/bin/bash
# BASELINE config 3 (64 independent 2^20-row segments) on N GPUs of one node: one process per GPU (zkm_amd/dist.py), four contexts per
# process, segments round-robin over the ranks, proofs gathered on rank 0 after the clock has stopped; rank 0 prints ONE JSON line.
#   tools/run_config3.sh [N=8] [SEGMENTS=64] [PORT=29500]
# N = 1 needs no launcher.  Refuses (with a message) when N does not match the visible GPUs.
N=${1:-8}
SEGMENTS=${2:-64}
PORT=${3:-29500}
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}      # RCCL on this driver stack: dmabuf IPC
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}                      # streams of a process spread over 16 hardware queues
if [ "$N" -le 1 ]; then
    exec python bench.py --gpus 1 --segments "$SEGMENTS" --steps "$SEGMENTS" --warmup 1 --no-extras
fi
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
    bench.py --gpus "$N" --segments "$SEGMENTS" --steps $(( (SEGMENTS + N - 1) / N )) --warmup 1 --no-extras
