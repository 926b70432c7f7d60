#!/bin/bash
# round 4, first look at the rolling-epilogue GEMM (variant bit 13): race screen, same-process A/B per shape, step A/B
TAG=${1:-r4a}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== race screen"; timeout 600 python tools/gemm_race_screen.py 8196 1024 10 2>&1 | tee gpurun_out/${TAG}_race.jsonl | cut -c1-230
echo "=== gemm_bench 4 vs 8196"; GEMM_BENCH_VARIANTS=4,8196 GEMM_BENCH_NO_TN=1 timeout 600 tools/gemm_bench 1024 3 2>&1 | cut -c1-200 | tee gpurun_out/${TAG}_gemm_bench.jsonl
echo "=== k64 tests under the variant"; ANTMMF_GEMM_VARIANT=8196 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 -k "gemm_k64_persistent or uneven or linearity or persistent_ring" 2>&1 | tail -5
echo "=== bench default"; timeout 600 python bench.py --no-cpu-baseline --gemm-table gpurun_out/${TAG}_gemm_table_base.txt > gpurun_out/${TAG}_bench_base.json 2> gpurun_out/${TAG}_bench_base.err; cut -c1-600 gpurun_out/${TAG}_bench_base.json
echo "=== bench variant 8196"; ANTMMF_GEMM_VARIANT=8196 timeout 600 python bench.py --no-cpu-baseline --gemm-table gpurun_out/${TAG}_gemm_table_roll.txt > gpurun_out/${TAG}_bench_roll.json 2> gpurun_out/${TAG}_bench_roll.err; cut -c1-600 gpurun_out/${TAG}_bench_roll.json
head -30 gpurun_out/${TAG}_gemm_table_roll.txt
