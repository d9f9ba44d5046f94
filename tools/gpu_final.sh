#!/bin/bash
# Round-end evidence on ONE box: the GPU parity suite, smoke(), counter + trace profiles of the scans (-> bench.SCAN_PMC_FILE = profiles/r06_scan_pmc.json,
# stamped with cad_version()), the whole-step kernel trace, the default bench line and the other configurations' bench lines.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
# ("quick": the kernel-level files only -- the whole suite takes 9 GPU-minutes, the driver runs it again at round end)
if [ "$1" = quick ]; then SUITE="tests/test_kernels.py tests/test_proj.py tests/test_fp8.py tests/test_equivariance.py"; else SUITE=tests; fi
timeout 1800 python -m pytest $SUITE -m gpu -q > gpurun_out/pytest_gpu_final.log 2>&1; grep -n "passed\|failed" gpurun_out/pytest_gpu_final.log | tail -2; grep -n "^FAILED\|^ERROR" gpurun_out/pytest_gpu_final.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
rm -rf gpurun_out/prof; timeout 900 bash tools/prof_scan.sh > gpurun_out/prof_scan.log 2>&1
python tools/make_scan_pmc_json.py gpurun_out/prof gpurun_out/scan_pmc.json 2>&1 | tail -1 | cut -c1-400
python tools/summarize_prof.py gpurun_out/prof > gpurun_out/prof_summary.txt 2>&1; grep -n "scan_bwd_kernel\|scan_fwd_kernel" gpurun_out/prof_summary.txt | head -2 | cut -c1-260
timeout 600 bash tools/prof_step.sh 2>&1 | tail -1 | cut -c1-200
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_step/trace/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    NS = 4  # bench.py --steps 2 --warmup 1 + its one extra untimed step (all kernel families timed)
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out = [f"total kernel time per step: {tot / NS / 1e6:.1f} ms ({NS} steps traced incl. warm-up and the extra instrumented step; "
           "model initialisation is in the totals: fills, normal_)"]
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
        out.append(f"{float(r['TotalDurationNs']) / NS / 1e6:8.2f} ms/step {float(r['Percentage']):6.2f}% calls/step={int(r['Calls']) / NS:7.1f} avg_us={float(r['AverageNs']) / 1e3:9.1f}  {r['Name'][:150]}")
    open("gpurun_out/step_trace_final.txt", "w").write("\n".join(out) + "\n")
    print("\n".join(out[:14]))
PY
# (round 6: the dB / dC fold runs on a second stream next to the scan backward -- the plain sum above counts its duration in full; what the GPU
# was busy for, and how much of every kernel ran alone, is the interval union)
python tools/trace_overlap.py gpurun_out/prof_step/trace 4 > gpurun_out/step_trace_overlap_final.txt 2>&1; head -6 gpurun_out/step_trace_overlap_final.txt | cut -c1-220
cp gpurun_out/scan_pmc.json profiles/r06_scan_pmc.json  # (on the box: the bench line below quotes the counters taken on THIS build)
timeout 400 python bench.py > gpurun_out/bench_final.log 2>gpurun_out/bench_final.err; tail -1 gpurun_out/bench_final.log | cut -c1-400
timeout 400 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-floor > gpurun_out/bench_final_20steps.log 2>/dev/null; tail -1 gpurun_out/bench_final_20steps.log | cut -c1-200
if [ "$1" = all ]; then
timeout 300 python bench.py --model ph --seqlen 1024 --batch 128 --cpu-sample 0 --no-floor > gpurun_out/bench_c2_ph_L1024_b128.log 2>/dev/null; tail -1 gpurun_out/bench_c2_ph_L1024_b128.log | cut -c1-200
timeout 300 python bench.py --model ph --cpu-sample 0 > gpurun_out/bench_ph_L131072.log 2>/dev/null; tail -1 gpurun_out/bench_ph_L131072.log | cut -c1-200
timeout 400 python bench.py --d-model 512 --seqlen 262144 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/bench_c4shape_bf16_1gpu.log 2>/dev/null; tail -1 gpurun_out/bench_c4shape_bf16_1gpu.log | cut -c1-200
timeout 400 python bench.py --d-model 512 --seqlen 262144 --steps 3 --warmup 1 --cpu-sample 0 --fp8-proj > gpurun_out/bench_c4shape_fp8_1gpu.log 2>/dev/null; tail -1 gpurun_out/bench_c4shape_fp8_1gpu.log | cut -c1-200
CADUCEUS_AMD_LIB_OUT_X_PROJ_D512=1 timeout 400 python bench.py --d-model 512 --seqlen 262144 --steps 3 --warmup 1 --cpu-sample 0 --no-floor > gpurun_out/bench_c4shape_bf16_lib_out_x_proj.log 2>/dev/null; tail -1 gpurun_out/bench_c4shape_bf16_lib_out_x_proj.log | cut -c1-200
timeout 900 bash tools/prof_step_c4.sh 2>&1 | grep -v "^W2026" | cut -c1-200 | head -8
CADUCEUS_DP_FORCE_COLLECTIVE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --cpu-sample 0 --no-floor > gpurun_out/bench_torchrun_1rank_rccl.log 2>/dev/null; tail -1 gpurun_out/bench_torchrun_1rank_rccl.log | cut -c1-200
timeout 300 python bench.py --global-batch 8 --steps 2 --warmup 1 --cpu-sample 0 --no-floor > gpurun_out/bench_global_batch8_1gpu.log 2>/dev/null; tail -1 gpurun_out/bench_global_batch8_1gpu.log | cut -c1-200
fi
if [ "$1" = all ]; then
# the fp32 path: parity, cad_gemm_f32 against hipBLASLt, step traces without a library GEMM
bash tools/gpu_f32.sh > gpurun_out/gpu_f32.log 2>&1; grep -n "passed\|library GEMM" gpurun_out/gpu_f32.log | cut -c1-160
# the instruction prices bench.py's roofline.valu uses (VALU_PRICE_NS), re-measured on this box
(cd /tmp && timeout 300 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench "$OLDPWD/tools/ubench/ubench.hip" 2>/dev/null && timeout 300 /tmp/ubench) > gpurun_out/ubench_gfx950.log 2>&1; grep -c "waves/SIMD=2" gpurun_out/ubench_gfx950.log
fi
