#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 bash tools/ab_scan.sh 2 default wi1 wi2 wi4 wi7 > /dev/null 2>&1; grep bwd2 gpurun_out/ab_scan.log
