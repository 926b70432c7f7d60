#!/bin/bash
# rolling-epilogue ablations: product (4 | 16384 = burst epilogue everywhere), rolling forced (8196), rolling without stores (+32768), rolling with the hook inside the MFMA block (+65536)
TAG=${1:-r4b}
mkdir -p gpurun_out
GEMM_BENCH_VARIANTS=${2:-16388,8196,40964,73732} GEMM_BENCH_NO_TN=1 timeout 600 tools/gemm_bench 1024 3 2>&1 | cut -c1-200 | tee gpurun_out/${TAG}_gemm_bench.jsonl
