#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 bash tools/ab_train_scan.sh 3 default lsreg revonly noswz > /dev/null 2>&1; cat gpurun_out/ab_train_scan.log
