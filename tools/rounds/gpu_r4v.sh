#!/bin/bash
# round 4, last validation of the final code: whole GPU suite, smoke(), default bench
TAG=${1:-r4v}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== pytest -m gpu"; python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log; tail -4 gpurun_out/${TAG}_pytest_gpu.log
echo "=== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | cut -c1-100
echo "=== bench"; timeout 900 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cut -c1-200 gpurun_out/${TAG}_bench.json
