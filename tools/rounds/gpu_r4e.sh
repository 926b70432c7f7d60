#!/bin/bash
# round 4: validation of the new defaults -- race screen, whole GPU suite, step bench with the burst-epilogue A/B
TAG=${1:-r4e}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== race screen (default dispatch)"; timeout 600 python tools/gemm_race_screen.py 4 1024 10 2>&1 | tee gpurun_out/${TAG}_race.jsonl | cut -c1-200 | tail -4
echo "=== pytest -m gpu"; python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log; tail -15 gpurun_out/${TAG}_pytest_gpu.log
echo "=== bench default"; timeout 600 python bench.py --no-cpu-baseline --gemm-table gpurun_out/${TAG}_gemm_table.txt > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cut -c1-300 gpurun_out/${TAG}_bench.json
echo "=== bench burst epilogue (variant 16388)"; ANTMMF_GEMM_VARIANT=16388 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench_burst.json 2> gpurun_out/${TAG}_bench_burst.err; cut -c1-300 gpurun_out/${TAG}_bench_burst.json
echo "=== bench default again"; timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench2.json 2> gpurun_out/${TAG}_bench2.err; cut -c1-300 gpurun_out/${TAG}_bench2.json
