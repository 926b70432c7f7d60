#!/bin/bash
# round 5, sixth call: the two new real-width cases; cells for up to 4 per workgroup (lab bit 26: leftover rounds of 6 - 8 tiles per chunk, J = 3072 / 4096) vs the product rule
TAG=${1:-r5f}
mkdir -p gpurun_out; export TMPDIR=/tmp
export ANTMMF_REAL_WIDTH_OUT=$PWD/gpurun_out/${TAG}_real_width.jsonl
timeout 900 python -m pytest tests/test_real_width_gpu.py -m gpu -q --timeout 900 -k "b16 or vtp8t" 2>&1 | tail -3
cut -c1-700 $ANTMMF_REAL_WIDTH_OUT
echo "=== product rule (4) vs up to 4 cells per workgroup (67108868)"
GEMM_BENCH_VARIANTS=4,67108868 GEMM_BENCH_NO_TN=1 timeout 600 tools/gemm_bench 1024 3 2>&1 | tee gpurun_out/${TAG}_gemm_cells_more_ab.jsonl | cut -c1-150
echo "=== the same on the text tower's rows (308 row tiles)"
GEMM_BENCH_VARIANTS=4,67108868 GEMM_BENCH_NO_TN=1 GEMM_BENCH_TOKENS=78848 timeout 600 tools/gemm_bench 1024 3 2>&1 | tee gpurun_out/${TAG}_gemm_cells_more_text_ab.jsonl | cut -c1-150
