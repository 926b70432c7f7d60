#!/bin/bash
TAG=${1:-q}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 -x -k "gemm" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest.log
timeout 300 python tools/kernel_bench.py > gpurun_out/${TAG}_kernel_bench.jsonl 2> gpurun_out/${TAG}_kernel_bench.err; grep -E "gemm" gpurun_out/${TAG}_kernel_bench.jsonl | cut -c1-130
