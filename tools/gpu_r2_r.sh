#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 bash tools/ab_train_scan.sh 1 default w64 w128 w512 w1024 w2048 w4096 default > /dev/null 2>&1; cat gpurun_out/ab_train_scan.log
