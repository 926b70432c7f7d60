#!/bin/bash
TAG=${1:-r3d}
mkdir -p gpurun_out
echo "--- normal"; GEMM_BENCH_VARIANTS=4 GEMM_BENCH_NO_TN=1 timeout 300 tools/gemm_bench 1024 3 2>&1 | grep '"fc2"\|"out"\|dgrad_fc1\|dgrad_out' | cut -c1-220 | tee gpurun_out/${TAG}_res_normal.jsonl
echo "--- residual ld = 0 (cache-resident residual)"; GEMM_BENCH_RES_LD0=1 GEMM_BENCH_VARIANTS=4 GEMM_BENCH_NO_TN=1 timeout 300 tools/gemm_bench 1024 3 2>&1 | grep '"fc2"\|"out"' | cut -c1-220 | tee gpurun_out/${TAG}_res_ld0.jsonl
