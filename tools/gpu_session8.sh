#!/bin/bash
# round-3 GPU session 8: B/C tile image (LDS-DMA) vs staged tiles: parity on the device, same-box A/B through the env switch
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py tests/test_equivariance.py -m gpu -x -q 2>&1 | tail -2
: > gpurun_out/ab_tiles.log
for r in 1 2 3; do
  for t in 0 1; do
    CADUCEUS_AMD_BC_TILES=$t timeout 200 python tools/layer_bench.py 2>/dev/null | grep layer_ms | sed "s/^/tiles=$t /" | tee -a gpurun_out/ab_tiles.log | cut -c1-230
  done
done
for t in 0 1; do
  CADUCEUS_AMD_BC_TILES=$t timeout 300 python bench.py --cpu-sample 0 > gpurun_out/bench_tiles$t.log 2>gpurun_out/bench_tiles$t.err; tail -1 gpurun_out/bench_tiles$t.log | cut -c1-200
done
CADUCEUS_AMD_BC_TILES=1 timeout 300 python bench.py --model ph --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/bench_ph_tiles1.log 2>&1; tail -1 gpurun_out/bench_ph_tiles1.log | cut -c1-200
