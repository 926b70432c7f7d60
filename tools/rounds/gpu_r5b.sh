#!/bin/bash
# round 5, second call: the product / lab split + the in-kernel cell tail on hardware -- full -m gpu suite, cells vs no cells per shape (same process), patch-shape A/B, step bench
TAG=${1:-r5b}
mkdir -p gpurun_out; export TMPDIR=/tmp
export ANTMMF_REAL_WIDTH_OUT=$PWD/gpurun_out/${TAG}_real_width.jsonl
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
tail -6 gpurun_out/${TAG}_pytest_gpu.log
echo "=== cells (4) vs no cells (33554436), per shape, same process"
GEMM_BENCH_VARIANTS=33554436,4 GEMM_BENCH_NO_TN=1 timeout 600 tools/gemm_bench 1024 3 2>&1 | tee gpurun_out/${TAG}_gemm_cells_ab.jsonl | cut -c1-150
echo "=== patch shape (band height of the XCD tile patch): 4 rows (product) / 8 / 16 / 2"
for r in 1 33 65 97 1; do echo "--- ANTMMF_GEMM_RASTER=$r"; ANTMMF_GEMM_RASTER=$r GEMM_BENCH_VARIANTS=4 GEMM_BENCH_NO_TN=1 timeout 300 tools/gemm_bench 1024 2 2>&1 | sed "s/^/raster $r /" | tee -a gpurun_out/${TAG}_gemm_patch_shape_ab.txt | cut -c1-130; done
echo "=== bench"; timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --gemm-table gpurun_out/${TAG}_gemm_table_l14.txt > gpurun_out/${TAG}_bench_l14.json 2> gpurun_out/${TAG}_bench_l14.err; tail -2 gpurun_out/${TAG}_bench_l14.err; cut -c1-400 gpurun_out/${TAG}_bench_l14.json
