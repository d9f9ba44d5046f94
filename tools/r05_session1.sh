#!/bin/bash
# GPU session 1 of round 5: parity of the changed scans, same-box A/B against the round-4 library, one bench line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "from caduceus_amd import _lib; print(_lib.version())" > gpurun_out/s1_version.txt 2>&1
timeout 900 python -m pytest tests/test_kernels.py tests/test_proj.py tests/test_equivariance.py -m gpu -x -q -k "scan or lean or production_scans or mirror or mixer_layer" > gpurun_out/s1_pytest.log 2>&1
tail -3 gpurun_out/s1_pytest.log
bash tools/ab_layer.sh 3 base default > gpurun_out/s1_ab.log 2>&1
cut -c1-400 gpurun_out/s1_ab.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/s1_bench.log 2>&1
tail -c 3000 gpurun_out/s1_bench.log
