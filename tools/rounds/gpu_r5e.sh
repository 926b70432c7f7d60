#!/bin/bash
# round 5, fifth call: L2 warm-up in front of the cells' K loops -- same-process A/B (no cells / cells without warm-up / cells with warm-up), the cell tests, a short bench
TAG=${1:-r5e}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 900 -k "gemm" 2>&1 | tail -3
echo "=== no cells (33554436) / cells, no warm-up (134217732) / cells + warm-up (4)"
GEMM_BENCH_VARIANTS=33554436,134217732,4 GEMM_BENCH_NO_TN=1 timeout 600 tools/gemm_bench 1024 3 2>&1 | tee gpurun_out/${TAG}_gemm_cells_warmup_ab.jsonl | cut -c1-150
echo "=== bench"; timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --gemm-table gpurun_out/${TAG}_gemm_table_l14.txt > gpurun_out/${TAG}_bench_l14.json 2> gpurun_out/${TAG}_bench_l14.err; tail -2 gpurun_out/${TAG}_bench_l14.err; cut -c1-300 gpurun_out/${TAG}_bench_l14.json
