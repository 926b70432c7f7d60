#!/bin/bash
TAG=${1:-r3c}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_e2e_gpu.py -m gpu -q --timeout 900 -x -k "full_size or registry" > gpurun_out/${TAG}_pytest_new.log 2>&1; tail -5 gpurun_out/${TAG}_pytest_new.log
GEMM_BENCH_HIPBLASLT=1 GEMM_BENCH_VARIANTS=4 timeout 600 tools/gemm_bench 1024 3 > gpurun_out/${TAG}_gemm_bench_hipblaslt_1024pairs.jsonl 2>&1; grep hipblaslt gpurun_out/${TAG}_gemm_bench_hipblaslt_1024pairs.jsonl | cut -c1-420
