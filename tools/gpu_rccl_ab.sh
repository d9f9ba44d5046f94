#!/bin/bash
# GPU box: the concurrent fold under a process group (1 rank, RCCL, forced collectives) and without one, same box
cd "$(dirname "$0")/.."
export CADUCEUS_AMD_VERBOSE=1
for e in "CADUCEUS_AMD_STREAM_FOLD=0" "CADUCEUS_AMD_STREAM_FOLD=1"; do
  echo "rccl $e"
  env $e CADUCEUS_DP_FORCE_COLLECTIVE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --steps 10 --warmup 3 --cpu-sample 0 --no-floor 2>/dev/null | grep -E "caduceus_amd:|metric" | cut -c1-220
done
for e in "CADUCEUS_AMD_STREAM_FOLD=0" "CADUCEUS_AMD_STREAM_FOLD=1"; do
  echo "plain $e"
  env $e python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-floor 2>/dev/null | grep -E "caduceus_amd:|metric" | cut -c1-220
done
