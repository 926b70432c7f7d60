#!/bin/bash
# round 4, last validation of the final code: race screen of the GEMM dispatch, whole GPU suite, smoke()
TAG=${1:-r4v}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== race screen (default dispatch)"; timeout 600 python tools/gemm_race_screen.py 4 1024 10 2>&1 | tee gpurun_out/${TAG}_race.jsonl | cut -c1-200 | tail -4
echo "=== pytest -m gpu"; python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log; tail -4 gpurun_out/${TAG}_pytest_gpu.log
echo "=== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
