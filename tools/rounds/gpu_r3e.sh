#!/bin/bash
TAG=${1:-r3e}
mkdir -p gpurun_out
GEMM_BENCH_VARIANTS=4,8196,24580,4100,2052 GEMM_BENCH_NO_TN=1 timeout 300 tools/gemm_bench 1024 3 2>&1 | grep '"fc2"\|"out"' | cut -c1-240 | tee gpurun_out/${TAG}_res_epilogue_variants.jsonl
